#!/usr/bin/env python3
"""The encoder phase alone (P3: set_conditional_embedding = the two StyleGAN_zxc generators on two streams) as a hipGraph: median replay time.
A/B switches travel in the environment (e.g. HAVATAR_CONV_KSPLIT_CUS).  python tools/bench_encoders.py [replays]"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from havatar_amd import synth
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = CfgNode.load_yaml(os.path.join(ROOT, "havatar_amd", "config", "hd_base.yml"))
cfg.models.StyleUnet.inp_size = 512
torch.manual_seed(0)
tr = Trainer(cfg, 1)
tr.requires_grad_(False)
synth.fill_state_dict(tr)
tr = tr.to(dev)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
front, left, right = [t(a) for a in synth.cond_images()]
pose = t(synth.frame_pose(0))[None]


def enc():
    with torch.no_grad():
        tr.model_coarse.set_conditional_embedding(front_render_cond=front, left_render_cond=left, right_render_cond=right,
                                                  latents=tr.latent_codes[0:1], cond_c=pose.view(1, -1))


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        enc()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    enc()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(30):
    g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts = sorted(ts)
print("encoder phase as a hipGraph: median %.3f ms, min %.3f, p90 %.3f (%d replays)" % (ts[len(ts) // 2], ts[0], ts[int(0.9 * len(ts))], n))
