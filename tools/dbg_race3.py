import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.render import RayMarcher
N = int(os.environ.get("NN", "512")); SF = int(os.environ.get("SF", "0")); P = int(os.environ.get("PERTURB", "0"))
sc = synth.scene(8, 8, "primary"); dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
rays = t(synth.camera_rays(N, N))[None]; bg = torch.ones(1, N*N, 3, device=dev)
def run():
    if P: rm.rng_counter = None
    return rm.render(rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, SF, perturb=bool(P))
ref = run()
tot = 0
for k in range(int(os.environ.get("TRIALS", "12"))):
    o = run(); torch.cuda.synchronize()
    bad = torch.zeros(N * N, dtype=torch.bool, device=dev)
    for i in range(7):
        if ref[i] is not None: bad |= ((ref[i] - o[i]).abs().reshape(N * N, -1).max(1)[0] > 0)
    idx = torch.nonzero(bad).reshape(-1).cpu().numpy()
    tot += idx.size
    if idx.size: print("trial", k, "bad rays", idx.size, "lanes", sorted(set((idx % 32).tolist())), "blocks", sorted(set((idx // 32).tolist()))[:6])
print("N", N, "S_f", SF, "perturb", P, "total differing over trials:", tot)
