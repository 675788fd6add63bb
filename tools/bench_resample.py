#!/usr/bin/env python3
"""Importance resampling between the two passes of the training forward at config 5's size (2 x 4096 rays, 64 coarse depths, 64 new
samples): hav_resample_depths (one launch) against the ATen statement of model/nerf_trainer.py:166-170 (utils/nerf_util.py::sample_pdf,
cat, sort), eager launches timed with HIP events on the current stream, and the bytes the kernel has to move."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from havatar_amd.native.train_ops import resample_depths
from havatar_amd.utils.nerf_util import sample_pdf

dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, S_c, S_f in ((8192, 64, 64), (8192, 64, 16), (262144, 64, 16)):
    z = torch.sort(torch.rand(n, S_c, device=dev) * 1.6 + 3.6, dim=-1)[0]
    w = torch.rand(n, S_c, device=dev) ** 8 * (torch.rand(n, S_c, device=dev) < 0.3)
    zeta = torch.rand(n, S_f, device=dev)

    def aten():
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        z_s = sample_pdf(z_mid, w[..., 1:-1], S_f, det=True).detach()      # det: no host draw inside the timed region
        return torch.sort(torch.cat((z[:, ::2], z_s), dim=-1), dim=-1)[0]

    def hip():
        return resample_depths(z, w, S_f, zeta)

    res = {}
    for name, fn in (("aten", aten), ("hip", hip)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps * 1e3
    nbytes = 4 * n * (2 * S_c + S_f + (S_c + 1) // 2 + S_f)
    print(f"n={n} S_c={S_c} S_f={S_f}: ATen statement {res['aten']:.1f} us, hav_resample_depths {res['hip']:.1f} us "
          f"({nbytes / 1e6:.1f} MB read + written: {nbytes / res['hip'] / 1e3:.0f} GB/s)")
