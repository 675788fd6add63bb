#!/usr/bin/env python3
"""HBM roofline of the two streaming ops at the BASELINE config-4 shapes (SWGAN_unet 512->1024): achieved GB/s =
algorithmic bytes (read + write) / kernel time (HIP events on the launch stream), against 8 TB/s.

Every timed launch works on its OWN input/output buffers, and one timed sequence touches >= 1 GiB of distinct memory (4x the
256 MiB Infinity Cache), so no launch can be served from the die-level cache that the previous one filled: what is measured is
DRAM bandwidth.  (Round 1 replayed 20 launches on one <= 201 MB buffer pair and reported up to 123 % of the HBM peak.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native import fused, upfirdn2d

dev = torch.device("cuda:0")


FOOTPRINT = 1 << 30            # distinct bytes one timed sequence must touch


def timed(make, bytes_per_launch, reps=7):
    """make(i) -> a launch closure on buffer set i.  K = enough distinct sets to cover FOOTPRINT (8..64); the K launches are captured
    back to back in one hipGraph (no host gaps) and the graph is replayed `reps` times; median time per launch in ms."""
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
    return sorted(ts)[len(ts) // 2], K


if "--calib" in sys.argv:
    # known-byte streaming launch for calibrating FETCH_SIZE / WRITE_SIZE (larger than the 256 MiB Infinity Cache)
    x = torch.randn(4, 64, 512, 512, device=dev); b = torch.randn(64, device=dev); e = x.new_empty(0)
    for _ in range(6):
        fused.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5)
    torch.cuda.synchronize()
    print(json.dumps({"calib_kernel": "fba_vec_kernel<float, 4>", "read_bytes": 4 * x.numel() + 256, "write_bytes": 4 * x.numel()}))
    sys.exit(0)

BRIEF = "--brief" in sys.argv          # bench.py's extra.ops: the five cfg4-size rows only
rows = []
k1 = torch.tensor([1., 3., 3., 1.], device=dev)
k4 = (k1[None] * k1[:, None]); k4 = k4 / k4.sum()
haar = torch.tensor([[1., -1.], [1., -1.]], device=dev) / 2 ** 0.5
for name, shape in (("fused_bias_act [1,64,512,512]", (1, 64, 512, 512)), ("fused_bias_act [1,128,256,256]", (1, 128, 256, 256)),
                    ("fused_bias_act [1,512,64,64]", (1, 512, 64, 64)))[:1 if BRIEF else 3]:
    b = torch.randn(shape[1], device=dev)
    e = b.new_empty(0)
    n_el = 1
    for d in shape:
        n_el *= d

    def mk_fwd(i):
        x = torch.randn(shape, device=dev)
        return lambda: fused.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5)

    def mk_bwd(i):
        x, r = torch.randn(shape, device=dev), torch.randn(shape, device=dev)
        return lambda: fused.fused_bias_act(x, e, r, 3, 1, 0.2, 2 ** 0.5)
    ms, K = timed(mk_fwd, 8 * n_el)
    rows.append((name, ms, 8 * n_el + 4 * b.numel(), K))
    ms, K = timed(mk_bwd, 12 * n_el)
    rows.append((name + " grad=1", ms, 12 * n_el, K))
    torch.cuda.empty_cache()
# the blur behind the up-sampling StyledConv of SWGAN_unet's 512^2 level: conv_transpose2d 256^2 -> 513^2, Blur(pad = (1, 1)) -> 512^2
# (model/styleUnet.py:186-194; rounds 1-4 timed this row with pad (2, 1) -> 513^2, which is now the last row of the full table)
for name, shape, k, up, dn, pad in (("upfirdn2d blur k4 [64,513,513] pad (1,1)", (64, 513, 513, 1), k4, 1, 1, (1, 1, 1, 1)),
                                    ("upfirdn2d down2 k4 [64,513,513]", (64, 513, 513, 1), k4, 1, 2, (1, 1, 1, 1)),
                                    ("upfirdn2d up2 k4 [12,512,512]", (12, 512, 512, 1), k4 * 4, 2, 1, (2, 1, 2, 1)),
                                    ("upfirdn2d up2 haar [3,512,512]", (3, 512, 512, 1), haar, 2, 1, (1, 0, 1, 0)),
                                    ("upfirdn2d down2 haar [3,1024,1024]", (3, 1024, 1024, 1), haar, 1, 2, (0, 0, 0, 0)),
                                    ("upfirdn2d blur k4 [512,35,35]", (512, 35, 35, 1), k4, 1, 1, (2, 1, 2, 1)),
                                    ("upfirdn2d blur k4 [64,513,513] pad (2,1) -> 513^2", (64, 513, 513, 1), k4, 1, 1, (2, 1, 2, 1)))[:3 if BRIEF else 7]:
    y = upfirdn2d.upfirdn2d(torch.randn(shape, device=dev), k, up, up, dn, dn, *pad)
    n_in = 1
    for d in shape:
        n_in *= d
    by = 4 * (n_in + y.numel())

    def mk(i):
        x = torch.randn(shape, device=dev)
        return lambda: upfirdn2d.upfirdn2d(x, k, up, up, dn, dn, *pad)         # (each call allocates its own output)
    ms, K = timed(mk, by)
    rows.append((name, ms, by, K))
    torch.cuda.empty_cache()
# the Haar analysis / synthesis of SWGAN_unet's skip path as one pass each (hav_haar_dwt / hav_haar_idwt): four bands in one launch --
# compare with 4 x the per-band upfirdn2d rows above (+ a cat / three adds)
bank = torch.stack([torch.tensor(q, device=dev) for q in ([[.5, .5], [.5, .5]], [[-.5, -.5], [.5, .5]], [[-.5, .5], [-.5, .5]], [[.5, -.5], [-.5, .5]])])
for name, shape, inv in (("haar analysis, 4 bands, one pass [3,1024,1024]", (1, 3, 1024, 1024), False),
                         ("haar synthesis, 4 bands, one pass [12,512,512]", (1, 12, 512, 512), True))[:0 if BRIEF else 2]:
    n_in = 3 * 1024 * 1024

    def mkh(i):
        x = torch.randn(shape, device=dev)
        return lambda: fused.haar(x, bank, inverse=inv)
    ms, K = timed(mkh, 8 * n_in)
    rows.append((name, ms, 8 * n_in, K))
    torch.cuda.empty_cache()
print("# every launch on its own buffers, >= 1 GiB distinct per timed sequence (K launches): DRAM bandwidth, not Infinity Cache")
for name, ms, by, K in rows:
    print("%-40s %8.1f us  %7.1f MB  %7.0f GB/s  %5.1f %% of 8 TB/s   (K=%d)" % (name, ms * 1e3, by / 1e6, by / ms / 1e6, 100 * by / ms / 1e6 / 8000, K))
print(json.dumps([{"op": n, "us": round(ms * 1e3, 2), "bytes": by, "GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                   "distinct_buffer_sets": K} for n, ms, by, K in rows]))
