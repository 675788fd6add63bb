import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.render import RayMarcher
H = W = 512
sc = synth.scene(8, 8, "primary")
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
rays = t(synth.camera_rays(H, W))[None]
bg = torch.ones(1, H * W, 3, device=dev)
S_f = int(os.environ.get("SF", "16"))
a = rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, S_f)
b = rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, S_f)
torch.cuda.synchronize()
d = (a[1] - b[1]).abs().reshape(-1).cpu().numpy()        # depth_coarse
idx = np.nonzero(d > 0)[0]
print("differing rays:", idx.size, "max", d.max())
if idx.size:
    blk = idx // 32
    ub, cnt = np.unique(blk, return_counts=True)
    print("blocks touched:", ub.size, "rays per touched block: min %d max %d mean %.1f" % (cnt.min(), cnt.max(), cnt.mean()))
    print("lane histogram (idx%32):", np.bincount(idx % 32, minlength=32))
    print("first blocks:", ub[:20])
    # feature channels too
    df = (a[0] - b[0]).abs().reshape(-1, 67).cpu().numpy()
    ch = (df > 0).sum(0)
    print("channels differing (count per channel):", ch[:8], "...", ch[60:])
