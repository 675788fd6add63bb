#!/usr/bin/env python3
"""EqualLinear(32 -> C) under autograd, forward + backward, B = 2 (the modulation layers of a training step): hav_equal_linear_* against
the ATen statement F.linear(x, W * scale, b * lr_mul); eager launches, HIP events, 200 repetitions."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from havatar_amd.native.train_ops import equal_linear

dev = torch.device("cuda:0")
for B, n_in, n_out in ((2, 32, 512), (2, 32, 64), (8, 512, 512)):
    x = torch.randn(B, n_in, device=dev, requires_grad=True)
    W = torch.randn(n_out, n_in, device=dev, requires_grad=True)
    b = torch.randn(n_out, device=dev, requires_grad=True)
    gy = torch.randn(B, n_out, device=dev)
    scale = 1 / math.sqrt(n_in)
    res = {}
    for name, fn in (("aten", lambda: torch.nn.functional.linear(x, W * scale, b * 1.0)), ("hip", lambda: equal_linear(x, W, b, scale, 1.0))):
        for _ in range(10):
            torch.autograd.grad(fn(), [x, W, b], gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            torch.autograd.grad(fn(), [x, W, b], gy)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 200 * 1e3
    print(f"B={B} in={n_in} out={n_out}: forward + backward, ATen statement {res['aten']:.1f} us, hav_equal_linear_* {res['hip']:.1f} us (eager, host-bound)")
