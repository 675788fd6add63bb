#!/usr/bin/env python3
"""Down-sampling ConvLayer (Blur -> 3x3 stride 2 -> bias + leaky-ReLU; model/styleUnet.py:326-368) at the encoders' shapes: the MIOpen
route (upfirdn2d + Im2d2Col + fp32 GEMM + fused_bias_act) against hav_upfirdn2d + hav_absmax + hav_conv3x3s2_split.  Each route is
captured as a hipGraph of 20 calls; prints microseconds per layer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from havatar_amd.model.styleUnet import ConvLayer

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, cin, cout, H in ((1, 256, 512, 128), (1, 512, 512, 64), (1, 64, 128, 256), (1, 128, 256, 128), (2, 256, 512, 128), (2, 512, 512, 64)):
    layer = ConvLayer(cin, cout, 3, downsample=True).to(dev).eval()
    x = torch.randn(B, cin, H, H, device=dev)
    res = {}
    for route in ("0", "1"):
        os.environ["HAVATAR_CONV_S2"] = route
        with torch.no_grad():
            for _ in range(3):
                y = layer(x)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                layer(x)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    y = layer(x)
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 20 * 1e3)
        res[route] = (float(np.median(ts)), y)
    err = (res["0"][1] - res["1"][1]).abs().max().item() / res["0"][1].abs().max().item()
    gf = 2 * 9 * cin * cout * (H // 2) ** 2 * B / 1e9
    print("B=%d %4d -> %4d @ %3d^2 -> %3d^2 (%.1f GFLOP): MIOpen route %6.1f us | split-fp16 route %6.1f us (%.0f TFLOP/s incl. blur, absmax) | rel. diff %.1e"
          % (B, cin, cout, H, H // 2, gf, res["0"][0], res["1"][0], gf / res["1"][0] * 1e3, err))
