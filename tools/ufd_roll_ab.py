#!/usr/bin/env python3
"""A/B of the upfirdn2d kernel families at the BASELINE config-4 sizes: the row-streaming kernels (per rows-per-segment choice; for the
decimating class they are not the default) against the strip / tiled kernels, bit identity of the results first, then time per launch with every launch on its own buffers (>= 1 GiB distinct
per timed sequence, as tools/bench_ops.py).  The choice is switched at run time through the library's lab hook hav_lab_upfirdn2d."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd import _lib
from havatar_amd.native import upfirdn2d, fused

dev = torch.device("cuda:0")
L = _lib.lib()
FOOTPRINT = 1 << 30


def timed(make, bytes_per_launch, reps=7):
    K = int(min(64, max(8, -(-FOOTPRINT // max(1, bytes_per_launch)))))
    fns = [make(i) for i in range(K)]
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / K)
    return sorted(ts)[len(ts) // 2]


k1 = torch.tensor([1., 3., 3., 1.], device=dev)
k4 = (k1[None] * k1[:, None]); k4 = k4 / k4.sum()
k3 = torch.randn(3, 3, device=dev)
haar = torch.tensor([[1., -1.], [1., -1.]], device=dev) / 2 ** 0.5
cases = [("blur k4 [64,512,512] pad 0,3 (aligned)", (64, 512, 512, 1), k4, 1, 1, (0, 3, 0, 3)),
         ("blur k4 [64,512,512] pad 2,1", (64, 512, 512, 1), k4, 1, 1, (2, 1, 2, 1)),
         ("blur k4 [64,513,513] pad 2,1", (64, 513, 513, 1), k4, 1, 1, (2, 1, 2, 1)),
         ("blur k4 [64,513,513] pad 1,1", (64, 513, 513, 1), k4, 1, 1, (1, 1, 1, 1)),
         ("blur k3 [64,512,512]", (64, 512, 512, 1), k3, 1, 1, (1, 1, 1, 1)),
         ("down2 k4 [64,513,513]", (64, 513, 513, 1), k4, 1, 2, (1, 1, 1, 1)),
         ("down2 k4 [64,512,512]", (64, 512, 512, 1), k4, 1, 2, (1, 1, 1, 1)),
         ("up2 k4 [12,512,512]", (12, 512, 512, 1), k4 * 4, 2, 1, (2, 1, 2, 1)),
         ("up2 haar [3,512,512]", (3, 512, 512, 1), haar, 2, 1, (1, 0, 1, 0)),
         ("blur k4 [128,257,257]", (128, 257, 257, 1), k4, 1, 1, (2, 1, 2, 1)),
         ("blur k4 [512,35,35]", (512, 35, 35, 1), k4, 1, 1, (2, 1, 2, 1)),
         ("up2 k4 [12,128,128]", (12, 128, 128, 1), k4 * 4, 2, 1, (2, 1, 2, 1))]
if len(sys.argv) > 1 and sys.argv[1].isdigit():
    cases = cases[:int(sys.argv[1])]
out = []
for name, shape, k, up, dn, pad in cases:
    x0 = torch.randn(shape, device=dev)
    L.hav_lab_upfirdn2d(0, 0)
    ref = upfirdn2d.upfirdn2d(x0, k, up, up, dn, dn, *pad)
    by = 4 * (x0.numel() + ref.numel())
    row = {"case": name, "MB": round(by / 1e6, 1)}

    def mk(i):
        x = torch.randn(shape, device=dev)
        return lambda: upfirdn2d.upfirdn2d(x, k, up, up, dn, dn, *pad)
    for mode, seg in ((0, 0), (2, 0), (2, 4), (2, 8), (2, 16), (2, 32)):
        L.hav_lab_upfirdn2d(mode, seg)
        y = upfirdn2d.upfirdn2d(x0, k, up, up, dn, dn, *pad)
        same = bool(torch.equal(y, ref))
        ms = timed(mk, by)
        tag = "tiled/strip" if mode == 0 else ("roll/auto" if seg == 0 else "roll/%d" % {4: 5, 8: 9, 16: 17, 32: 33}[seg])
        row[tag] = {"us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6), "bit_identical": same}
        torch.cuda.empty_cache()
    # ceiling: a streaming kernel of this library over the same number of bytes
    n = by // 8
    b = torch.randn(1, device=dev); e = b.new_empty(0)

    def mkc(i):
        xx = torch.randn(n, device=dev)
        return lambda: fused.fused_bias_act(xx, e, e, 3, 0, 0.2, 1.0)
    ms = timed(mkc, by)
    row["fused_bias_act same bytes"] = {"us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6)}
    torch.cuda.empty_cache()
    L.hav_lab_upfirdn2d(1, 0)
    out.append(row)
    print(json.dumps(row), flush=True)
