#!/usr/bin/env python3
"""Run-to-run differences of the fp16 + MX arithmetic mode (HAVATAR_LIB selects the build): the dense-layer hook on 2^18 queries, twice, and
6 launches of a 256 x 256 frame; how many results / rays differ from the first run and by how much."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.render import MLP_MODES, RayMarcher
dev = torch.device("cuda:0"); N = int(os.environ.get("SIZE", "256"))
mode = MLP_MODES[os.environ.get("MODE", "mx")]
sc = synth.scene(8, 8, "primary")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.mlp_mode = mode
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
g = torch.Generator(device="cpu").manual_seed(1)
for layer, K in ((1, 48), (2, 128)):
    x = (torch.rand(1 << 18, K, generator=g) * 2 - 1 if layer == 1 else torch.randn(1 << 18, K, generator=g).clamp_min(0)).to(dev)
    a = rm.mlp_layer(x, layer).clone()
    nd = 0
    for _ in range(4):
        b = rm.mlp_layer(x, layer)
        nd = max(nd, int((a != b).sum().item()))
    print("layer %d hook: %d of %d results differ between runs (max |diff| %.3e)" % (layer, nd, a.numel(), float((a - b).abs().max())))
rays = t(synth.camera_rays(N, N))[None]; bg = torch.ones(1, N * N, 3, device=dev)
args = (rays, bg, t(sc["inv_T"]), t(sc["vol"]), 64, 16)
for co in (False, True):
    try:
        ref = [o.clone() if o is not None else None for o in rm.render(*args, perturb=False, coarse_outputs=co)]
    except RuntimeError as e:
        print("coarse_outputs=%s: %s" % (co, str(e)[:60]))
        continue
    worst, wd = 0, 0.0
    for i in range(5):
        out = rm.render(*args, perturb=False, coarse_outputs=co); torch.cuda.synchronize()
        off = torch.zeros(N * N, dtype=torch.bool, device=dev)
        for a_, b_ in zip(ref, out):
            if a_ is not None:
                d = (a_ - b_).abs().reshape(N * N, -1).amax(1)
                off |= d > 0; wd = max(wd, float(d.max()))
        worst = max(worst, int(off.sum()))
    print("%s: up to %d of %d rays differ from the first launch (max |diff| %.3e)" % (rm.last_variant, worst, N * N, wd))
