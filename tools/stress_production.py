#!/usr/bin/env python3
"""400 launches of the 512x512 frame on each of the two production variants (fine maps only, with / without jitter), every
output compared bitwise with the first launch (docs/history/DESIGN_r1-r4.md 3.5)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from havatar_amd import _lib, synth
from havatar_amd.render import RayMarcher
dev = torch.device("cuda:0"); H = W = 512
sc = synth.scene(8, 8, "primary")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rays = t(synth.camera_rays(H, W))[None]; bg = torch.ones(1, H * W, 3, device=dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
args = (rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 16)
for perturb in (True, False):
    def go():
        if rm.rng_counter is not None: rm.rng_counter.zero_()
        return rm.render(*args, perturb=perturb, coarse_outputs=False)
    ref = [o.clone() if o is not None else None for o in go()]
    bad = 0
    for i in range(400):
        out = go(); torch.cuda.synchronize()
        bad += any((a is not None) and (not torch.equal(a, b)) for a, b in zip(ref, out))
    print(rm.variant(64, 16, perturb=perturb, coarse_outputs=False), ":", bad, "of 400 launches differ")
