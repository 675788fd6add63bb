#!/usr/bin/env python3
"""HBM roofline of the two streaming ops at the BASELINE config-4 shapes (SWGAN_unet 512->1024): achieved GB/s =
algorithmic bytes (read + write) / kernel time (HIP events on the launch stream), against 8 TB/s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native import fused, upfirdn2d

dev = torch.device("cuda:0")


def timed(fn, n=20, reps=9):
    """kernel time without host launch gaps: n back-to-back launches captured in a hipGraph, replayed `reps` times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[len(ts) // 2]


if "--calib" in sys.argv:
    # known-byte streaming launch for calibrating FETCH_SIZE / WRITE_SIZE (larger than the 256 MiB Infinity Cache)
    x = torch.randn(4, 64, 512, 512, device=dev); b = torch.randn(64, device=dev); e = x.new_empty(0)
    for _ in range(6):
        fused.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5)
    torch.cuda.synchronize()
    print(json.dumps({"calib_kernel": "fba_vec_kernel<float, 4>", "read_bytes": 4 * x.numel() + 256, "write_bytes": 4 * x.numel()}))
    sys.exit(0)

rows = []
k1 = torch.tensor([1., 3., 3., 1.], device=dev)
k4 = (k1[None] * k1[:, None]); k4 = k4 / k4.sum()
haar = torch.tensor([[1., -1.], [1., -1.]], device=dev) / 2 ** 0.5
for name, shape in (("fused_bias_act [1,64,512,512]", (1, 64, 512, 512)), ("fused_bias_act [1,128,256,256]", (1, 128, 256, 256)),
                    ("fused_bias_act [1,512,64,64]", (1, 512, 64, 64))):
    x = torch.randn(shape, device=dev); b = torch.randn(shape[1], device=dev); e = x.new_empty(0)
    ms = timed(lambda: fused.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5))
    by = 8 * x.numel() + 4 * b.numel()
    rows.append((name, ms, by))
    ms = timed(lambda: fused.fused_bias_act(x, e, x, 3, 1, 0.2, 2 ** 0.5))
    rows.append((name + " grad=1", ms, 12 * x.numel()))
for name, shape, k, up, dn, pad in (("upfirdn2d blur k4 [64,513,513]", (64, 513, 513, 1), k4, 1, 1, (2, 1, 2, 1)),
                                    ("upfirdn2d down2 k4 [64,513,513]", (64, 513, 513, 1), k4, 1, 2, (1, 1, 1, 1)),
                                    ("upfirdn2d up2 k4 [12,512,512]", (12, 512, 512, 1), k4 * 4, 2, 1, (2, 1, 2, 1)),
                                    ("upfirdn2d up2 haar [3,512,512]", (3, 512, 512, 1), haar, 2, 1, (1, 0, 1, 0)),
                                    ("upfirdn2d down2 haar [3,1024,1024]", (3, 1024, 1024, 1), haar, 1, 2, (0, 0, 0, 0)),
                                    ("upfirdn2d blur k4 [512,35,35]", (512, 35, 35, 1), k4, 1, 1, (2, 1, 2, 1))):
    x = torch.randn(shape, device=dev)
    y = upfirdn2d.upfirdn2d(x, k, up, up, dn, dn, *pad)
    ms = timed(lambda: upfirdn2d.upfirdn2d(x, k, up, up, dn, dn, *pad))
    rows.append((name, ms, 4 * (x.numel() + y.numel())))
for name, ms, by in rows:
    print("%-40s %8.1f us  %7.1f MB  %7.0f GB/s  %5.1f %% of 8 TB/s" % (name, ms * 1e3, by / 1e6, by / ms / 1e6, 100 * by / ms / 1e6 / 8000))
print(json.dumps([{"op": n, "us": round(ms * 1e3, 2), "bytes": by, "GBps": round(by / ms / 1e6, 1)} for n, ms, by in rows]))
