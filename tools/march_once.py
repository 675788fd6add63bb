#!/usr/bin/env python3
"""A few launches of the production ray-march kernel on one 512x512 frame (+ a streaming launch of known size that calibrates the
memory-side counters), and nothing else: the target of bench.py's live `rocprofv3 --pmc` passes (FETCH_SIZE / WRITE_SIZE in
separate passes, MI355X_MICROARCH.md HBM section) and of tools/profile.sh.  Prints one JSON line describing what ran."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from havatar_amd import synth
from havatar_amd.native import fused
from havatar_amd.render import RayMarcher

H = W = int(os.environ.get("SIZE", 512))
perturb = bool(int(os.environ.get("PERTURB", "1")))
n = int(os.environ.get("LAUNCHES", 3))
dev = torch.device("cuda:0")
sc = synth.scene(8, 8, "primary")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
rays = t(synth.camera_rays(H, W))[None]
bg = torch.ones(1, H * W, 3, device=dev)
inv_T, vol = t(sc["inv_T"]), t(sc["vol"])
for _ in range(n):
    rm.render(rays, bg, inv_T, vol, 64, 16, perturb=perturb, coarse_outputs=False)
torch.cuda.synchronize()
# known-byte streaming launch (larger than the 256 MiB Infinity Cache): 4 B read + 4 B written per element
x = torch.randn(4, 64, 512, 512, device=dev)
b = torch.randn(64, device=dev)
e = x.new_empty(0)
for _ in range(n):
    fused.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5)
torch.cuda.synchronize()
print(json.dumps({"march_kernel": rm.last_variant, "launches": n, "calib_kernel": "fba_vec_kernel<float, 4>",
                  "calib_read_bytes": 4 * x.numel() + 256, "calib_write_bytes": 4 * x.numel()}))
