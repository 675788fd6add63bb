#!/usr/bin/env python3
"""Per-phase cycle breakdown of the block march kernel (needs the -DHAV_PROFILE build, see tools/phase_profile.sh).
Prints mean cycles per 32-query tile per wave for each phase, for one 512^2 frame."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd import _lib, synth
from havatar_amd.render import RayMarcher

H = W = int(os.environ.get("SIZE", 512))
perturb = bool(int(os.environ.get("PERTURB", "1")))
dev = torch.device("cuda:0")
sc = synth.scene(H, W, "primary")
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
m = sc["mlp"]
t = lambda a: torch.from_numpy(a).to(dev)
rm.set_mlp(t(m["W1"]), t(m["b1"]), t(m["W2"]), t(m["b2"]), t(m["Wa"]), t(m["ba"]), t(m["Wf"]), t(m["bf"]), t(m["Wc"]), t(m["bc"]))
rm.set_triplane(t(sc["planes"]))
rays, bg, inv_T, vol = t(sc["rays"]), t(sc["bg"]), t(sc["inv_T"]), t(sc["vol"])
L = _lib.lib()
buf = (C.c_ulonglong * 24)()          # HAV_NPROF entries
for _ in range(2):
    rm.render(rays, bg, inv_T, vol, 64, 16, perturb=perturb, coarse_outputs=False)
L.hav_debug_read_prof(buf)
N = 5
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(N):
    rm.render(rays, bg, inv_T, vol, 64, 16, perturb=perturb, coarse_outputs=False)
ev1.record()
torch.cuda.synchronize()
assert L.hav_debug_read_prof(buf) == 0
tiles = H * W * 112 // 32
names = ["loop top (z, jitter)", "geometry + skinning taps", "plane gather (8 taps)", "positional encoding", "layer 1 MFMA + relu",
         "layer 2 MFMA + relu", "head dots", "compositing", "fc_rgbFeat + stores (per block)", "resampling (per block)", "kernel total / wave", "-"]
tot = sum(buf[i] for i in range(10))
print("kernel %.2f ms (instrumented), %d tiles, variant %s" % (ev0.elapsed_time(ev1) / N, tiles, rm.variant(64, 16, perturb=perturb, coarse_outputs=False)))
for i in range(10):
    print("  %-34s %9.0f cycles/tile  %5.1f %%" % (names[i], buf[i] / N / tiles, 100.0 * buf[i] / tot))
print("  %-34s %9.0f cycles/tile" % ("sum of phases", tot / N / tiles))
print("  wave lifetime: %.0f cycles mean (s_memtime ticks)" % (buf[10] / N / (8 * 256)))
