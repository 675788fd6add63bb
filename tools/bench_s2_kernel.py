#!/usr/bin/env python3
"""hav_conv3x3s2_split alone (no blur, with its absmax pass) at the encoders' two shapes, 20 calls per hipGraph: us per call.
HAVATAR_LIB selects an alternative library (timing experiments)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from havatar_amd.native import conv

dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
for B, cin, cout, H in ((1, 256, 512, 129), (1, 512, 512, 65), (2, 256, 512, 129)):
    x = torch.randn(B, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev)
    bias = torch.randn(cout, device=dev)
    pk = conv.pack(w, 1.0 / (cin * 9) ** 0.5)
    for _ in range(3):
        y = conv.conv3x3s2(x, pk, cout, 0, bias=bias, act=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            y = conv.conv3x3s2(x, pk, cout, 0, bias=bias, act=True)
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 20 * 1e3)
    out.append("B=%d %d->%d @%d: %.1f us" % (B, cin, cout, H, float(np.median(ts))))
print(os.path.basename(os.environ.get("HAVATAR_LIB", "(in-tree)")), " | ".join(out))
