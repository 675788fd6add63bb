#!/usr/bin/env python3
"""Run-to-run determinism stress of the march kernel (the symptom of the gfx950 MFMA operand hazard of docs/history/DESIGN_r1-r4.md 3.5 is a
fraction of rays that differ from launch to launch): N launches per variant, every output compared bitwise with the first launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import _lib, synth
from havatar_amd.render import RayMarcher

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
sc = synth.scene(8, 8, "primary")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rays = t(synth.camera_rays(H, W))[None]
bg = torch.ones(1, H * W, 3, device=dev)
names = ("rgb_coarse", "depth_coarse", "acc_coarse", "weights_max", "rgb_fine", "depth_fine", "acc_fine")
import itertools
for (mode_name, mode), fine, perturb, coarse in itertools.product(
        (("half", _lib.HAV_MLP_SPLIT_F16), ("split", _lib.HAV_MLP_SPLIT_BF16), ("f32", _lib.HAV_MLP_F32)), ("cache", "recompute"), (False, True), (True, False)):
    if mode_name == "f32" and (perturb or fine == "cache"):
        continue
    os.environ["HAV_FINE"] = fine
    if True:
        rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
        rm.mlp_mode = mode
        rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
        rm.set_triplane(t(sc["planes"]))
        args = (rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 16)
        def go():
            if rm.rng_counter is not None:
                rm.rng_counter.zero_()          # same jitter every launch
            return rm.render(*args, perturb=perturb, coarse_outputs=coarse)
        ref = [o.clone() if o is not None else None for o in go()]
        bad_runs, worst = 0, {}
        for i in range(N):
            out = go()
            torch.cuda.synchronize()
            any_bad = False
            for n, a, b in zip(names, ref, out):
                if a is None:
                    continue
                d = (a != b)
                if d.any():
                    any_bad = True
                    rows = d.reshape(H * W, -1).any(-1).nonzero().flatten()
                    e = (a - b).abs().max().item()
                    w = worst.setdefault(n, [0, 0.0, None])
                    w[0] = max(w[0], rows.numel()); w[1] = max(w[1], e)
                    if w[2] is None:
                        w[2] = np.bincount((rows % 32).cpu().numpy(), minlength=32).tolist()
            bad_runs += any_bad
        print("%-5s %-9s perturb=%-5s coarse_outputs=%-5s variant %-34s: %d of %d launches differ from the first" % (
            mode_name, fine, perturb, coarse, rm.variant(64, 16, perturb=perturb, coarse_outputs=coarse), bad_runs, N), {k: (v[0], "%.2e" % v[1], v[2]) for k, v in worst.items()})
