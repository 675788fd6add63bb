#!/usr/bin/env python3
"""hav_composite_{fwd,bwd} at config 5's pass sizes (8192 rays x 64 / 48 samples, 68 channels + density): time per call and HBM rate.
HAVATAR_COMPOSITE_BWD=direct selects the backward that reads its rows from memory (read once per process: run twice to compare)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native.train_ops import composite
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
out = {"mode": os.environ.get("HAVATAR_COMPOSITE_BWD", "staged")}
for S in (64, 48):
    n = 8192
    rf = torch.randn(n, S, 69, device=dev, generator=g, requires_grad=True)
    z = torch.sort(torch.rand(n, S, device=dev, generator=g) * 2 + 3, -1)[0]
    rd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
    noise = torch.randn(n, S, device=dev, generator=g) * 0.1
    bg = torch.rand(n, 3, device=dev, generator=g)
    ups = [torch.randn(n, 68, device=dev, generator=g), torch.randn(n, device=dev, generator=g), torch.randn(n, S, device=dev, generator=g), torch.randn(n, device=dev, generator=g)]

    def ev(fn, k=15):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(k):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    with torch.no_grad():
        tf = ev(lambda: composite(rf, z, rd, noise, bg, n_sigmoid=3))
    outs = composite(rf, z, rd, noise, bg, n_sigmoid=3)
    tb = ev(lambda: torch.autograd.grad(outs, rf, ups, retain_graph=True))
    gr, = torch.autograd.grad(outs, rf, ups, retain_graph=True)
    by = rf.numel() * 4
    out["S=%d" % S] = {"fwd_us": round(tf * 1e3, 1), "bwd_us": round(tb * 1e3, 1), "bwd_GBps": round(2 * by / tb / 1e6), "d_rf_checksum": float(gr.double().abs().sum())}
print(json.dumps(out))
