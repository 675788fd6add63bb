#!/usr/bin/env python3
"""hav_conv3x3_split vs MIOpen's fp32 convolution on the encoder's heavy 3x3 shapes (one frame, B=1): time and effective TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native import conv
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def timed(fn, n=10):
    """GPU time per call without host launch gaps: n calls captured in a hipGraph (as the frame runs them), replayed 7 times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(20):          # ~30 ms of the same work first: the clocks take a while to settle after an idle period
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / n)
    return sorted(ts)[len(ts) // 2]
def timed_eager(fn, n=10):
    """one call at a time, the GPU idle in between (events around each call)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for Cin, Cout, H in ((512, 512, 32), (1024, 512, 32), (512, 512, 64), (1024, 512, 64), (256, 256, 128), (512, 256, 128)):
    x = torch.randn(1, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device=dev)
    pk = conv.pack(w)
    t_ours = timed(lambda: conv.conv3x3(x, pk, Cout, bias=b))
    t_raw = timed(lambda: conv.conv3x3(x, pk, Cout, bias=b, autoscale=False))
    sm = torch.rand(1, Cin, device=dev) + 0.5
    dm = torch.rand(1, Cout, device=dev) + 0.5
    nz = torch.randn(1, 1, H, H, device=dev); nwt = torch.tensor([0.1], device=dev)
    t_mod = timed(lambda: conv.conv3x3(x, pk, Cout, s=sm, d=dm, noise=nz, noise_weight=nwt, bias=b))
    t_mi = timed(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    t_iso = timed_eager(lambda: conv.conv3x3(x, pk, Cout, bias=b, autoscale=False))
    t_mi_iso = timed_eager(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    fl = 2 * 9 * Cin * Cout * H * H
    print("%4d -> %4d @ %3d^2: split-fp16 %7.1f us (%5.0f TF/s eff; %6.1f us without the auto-scale pass)   MIOpen fp32 %7.1f us (%5.0f TF/s eff)" % (
        Cin, Cout, H, t_ours * 1e3, fl / t_ours / 1e9, t_raw * 1e3, t_mi * 1e3, fl / t_mi / 1e9))
    print("        one call at a time (GPU idle in between): split-fp16 %.1f us, MIOpen %.1f us;  as a StyledConv (modulation, demodulation, noise, bias, act fused): %.1f us" % (t_iso * 1e3, t_mi_iso * 1e3, t_mod * 1e3))

# ---- the up-sampling path: hav_gemm_split (transposed convolution before its scatter) and hav_upconv_finish, timed separately
import ctypes as C
from havatar_amd import _lib
from havatar_amd.model.styleUnet import make_kernel
L = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
print()
for Cin, Cout, H in ((512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256)):
    x = torch.randn(1, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev)
    s = 1.0 + 0.3 * torch.randn(1, Cin, device=dev)
    d = 0.5 + torch.rand(1, Cout, device=dev)
    fir = (make_kernel((1, 3, 3, 1)) * 4).to(dev)
    pk = conv.pack_upconv(w, 1.0 / (Cin * 9) ** 0.5)
    col = torch.empty(1, Cout * 9, H * H, device=dev)
    y = torch.empty(1, Cout, 2 * H, 2 * H, device=dev)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g_ms = timed(lambda: L.hav_gemm_split(p(col), p(x), p(pk), p(s), None, 1, Cout * 9, Cin, H * H, st()))
    f_ms = timed(lambda: L.hav_upconv_finish(p(y), p(col), p(fir), p(d), None, None, None, 0.2, 2 ** 0.5, 1, 0, 1, Cout, H, H, st()))
    flop = 2.0 * 9 * Cout * Cin * H * H
    byt = 4.0 * (9 * Cout * H * H + 4 * Cout * H * H)
    wt = (w * (1.0 / (Cin * 9) ** 0.5)).transpose(0, 1).contiguous()
    m_ms = timed(lambda: torch.nn.functional.conv_transpose2d(x * s.view(1, Cin, 1, 1), wt, stride=2))
    print("up-sampling %4d -> %4d @ %3d^2 -> %3d^2: gemm %6.1f us (%4.0f TF/s eff)  scatter + FIR + epilogue %6.1f us (%4.0f GB/s)   MIOpen conv_transpose2d alone %6.1f us" %
          (Cin, Cout, H, 2 * H, 1e3 * g_ms, flop / g_ms / 1e9, 1e3 * f_ms, byt / f_ms / 1e6, 1e3 * m_ms))
