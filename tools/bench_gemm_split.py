#!/usr/bin/env python3
"""hav_gemm_split (the transposed-convolution product of the up-sampling StyledConv, [9 Cout x Cin] . [Cin x HW]) on the layer shapes of
the two generators and of SWGAN_unet: time per launch inside a hipGraph and executed fp16-MFMA rate (3 products per fp32 product)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd import _lib
from havatar_amd.native import conv
dev = torch.device("cuda:0")
L = _lib.lib()
_p = lambda t: None if t is None else __import__("ctypes").c_void_p(t.data_ptr())


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[len(ts) // 2]


rows = []
for Cin, Cout, H in ((512, 512, 16), (512, 512, 32), (512, 512, 64), (512, 256, 128), (256, 128, 256), (128, 64, 512)):
    x = torch.randn(1, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    s = torch.rand(1, Cin, device=dev) + 0.5
    if not conv.upconv_eligible(x, w):
        continue
    packed = conv.pack_upconv(w)
    col = torch.empty(1, Cout * 9, H * H, device=dev)
    amax = conv.absmax(x)
    st = lambda: __import__("ctypes").c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        _lib.check(L.hav_gemm_split(_p(col), _p(x), _p(packed), _p(s), _p(amax), 1, Cout * 9, Cin, H * H, st()), "hav_gemm_split")
    ms = timed(run)
    ref = torch.einsum("mk,kn->mn", w.permute(0, 2, 3, 1).reshape(Cout * 9, Cin).double(), (x[0].double() * s[0].double()[:, None, None]).reshape(Cin, -1))
    err = float((col[0].double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * Cout * 9 * Cin * H * H
    rows.append({"Cin": Cin, "Cout": Cout, "H": H, "us": round(ms * 1e3, 1), "fp32_equiv_TFLOPs": round(fl / ms / 1e9, 1), "fp16_mfma_TFLOPs": round(3 * fl / ms / 1e9, 1),
                 "col_MB": round(col.numel() * 4 / 1e6, 1), "rel_err": err})
    print(json.dumps(rows[-1]), flush=True)
