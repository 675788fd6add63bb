import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.render import RayMarcher
N = 256
sc = synth.scene(8, 8, "primary"); dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
rays = t(synth.camera_rays(N, N))[None]; bg = torch.ones(1, N*N, 3, device=dev)
outs = [rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 0) for _ in range(3)]
torch.cuda.synchronize()
print("ablate", os.environ.get("HAV_ABLATE", "0"), "differing rays", [int(((outs[0][i]-outs[k][i]).abs().reshape(N*N, -1).max(1)[0] > 0).sum()) for k in (1, 2) for i in (0, 1)])
