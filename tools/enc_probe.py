import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = CfgNode.load_yaml(os.path.join(ROOT, "havatar_amd", "config", "hd_base.yml"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(cfg, 1); tr.requires_grad_(False); synth.fill_state_dict(tr); tr = tr.to(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
front, left, right = [t(a) for a in synth.cond_images()]
T = t(synth.inv_head_T())[None]
def enc():
    with torch.no_grad():
        tr.model_coarse.set_conditional_embedding(front_render_cond=front, left_render_cond=left, right_render_cond=right, latents=tr.latent_codes[0:1], cond_c=T.view(1, -1))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("default           : %.3f ms" % timeit(enc))
torch.backends.cudnn.benchmark = True
print("cudnn.benchmark   : %.3f ms" % timeit(enc))
ref = tr.model_coarse.triPlane_embeddings.clone()
# hipGraph capture of the encoder pass
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): enc()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        enc()
    print("graph replay      : %.3f ms" % timeit(g.replay))
    print("graph == eager    :", torch.equal(ref, tr.model_coarse.triPlane_embeddings))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
# kernel census of one eager pass
from torch.profiler import profile, ProfilerActivity
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    enc(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
