#!/usr/bin/env python3
"""hav_torgb on the SWGAN_unet(512 -> 1024) level shapes: time per call (20 calls on distinct buffers inside a hipGraph) and achieved GB/s
of algorithmic bytes (x read once + skip read + out written), against the unfused ATen statement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native import fused
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
def timed(fns):
    for f in fns[:2]: f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fns[0]()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / len(fns))
    return sorted(ts)[len(ts) // 2] * 1e3
for Cin, H in ((512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (64, 512)):
    n = max(2, min(20, int(1.2e9 / (Cin * H * H * 4))))          # >= 1 GiB of distinct x buffers where it fits (beyond the Infinity Cache)
    xs = [torch.randn(1, Cin, H, H, device=dev) for _ in range(n)]
    W = torch.randn(12, Cin, device=dev); s = 1 + 0.3 * torch.randn(1, Cin, device=dev); b = torch.randn(12, device=dev)
    skip = torch.randn(1, 12, H, H, device=dev)
    scale = Cin ** -0.5
    t_f = timed([lambda x=x: fused.torgb(x, W, s, b, skip, scale) for x in xs])
    w4 = (W * scale).view(12, Cin, 1, 1)
    t_a = timed([lambda x=x: torch.nn.functional.conv2d(x * s.view(1, Cin, 1, 1), w4) + b.view(1, 12, 1, 1) + skip for x in xs])
    byt = 4 * (Cin * H * H + 2 * 12 * H * H)
    print("ToRGB %4d -> 12 @ %4d^2: hav_torgb %6.1f us (%5.0f GB/s)   ATen statement %6.1f us" % (Cin, H, t_f, byt / t_f / 1e3, t_a))
