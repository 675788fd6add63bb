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
names = ["rgb_c", "depth_c", "acc_c", "wmax", "rgb_f", "depth_f", "acc_f"]
for trial in range(3):
    full = rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 16)
    full2 = rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 16)
    y0, y1 = 200, 203
    part = rm.render(rays[:, y0 * W:y1 * W].contiguous(), bg[:, y0 * W:y1 * W].contiguous(), t(sc["inv_T"]), t(sc["vol"]), 64, 16)
    torch.cuda.synchronize()
    for n, f, f2, p in zip(names, full, full2, part):
        d = (f[:, y0 * W:y1 * W] - p).abs()
        d2 = (f - f2).abs()
        print(trial, n, "full-vs-part: max %.3e nnz %d | full-vs-full: max %.3e nnz %d" % (d.max().item(), int((d > 0).sum()), d2.max().item(), int((d2 > 0).sum())))
