#!/usr/bin/env python3
"""Weight gradient of the 3x3 convolution: hav_conv3x3_wgrad (split-fp16 MFMA) vs ATen / MIOpen's fp32 route, training shapes (B = 2)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd.native import conv

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n)
    return float(np.median(ts))


for Cin, Cout, H in ((512, 512, 32), (1024, 512, 32), (512, 512, 64), (1024, 512, 64), (256, 256, 128), (512, 256, 128)):
    B = 2
    x = torch.randn(B, Cin, H, H, device=dev)
    go = torch.randn(B, Cout, H, H, device=dev) * 1e-6
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    flop = 2.0 * B * H * H * 9 * Cin * Cout
    t_h = timed(lambda: conv.wgrad3x3(go, x))
    t_m = timed(lambda: torch.ops.aten.convolution_backward(go, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    print("%4d -> %4d @ %3d^2 (B=2): split-fp16 wgrad %7.1f us (%4.0f TF/s eff)   ATen/MIOpen fp32 %7.1f us (%4.0f TF/s eff)" %
          (Cin, Cout, H, 1e3 * t_h, flop / t_h / 1e9, 1e3 * t_m, flop / t_m / 1e9))
