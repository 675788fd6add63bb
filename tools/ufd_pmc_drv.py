import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd import _lib
from havatar_amd.native import upfirdn2d, fused
dev = torch.device("cuda:0"); L = _lib.lib()
k1 = torch.tensor([1., 3., 3., 1.], device=dev); k4 = k1[None] * k1[:, None]; k4 = k4 / k4.sum()
case = sys.argv[1] if len(sys.argv) > 1 else "blur"
up, dn, pad, shape = {"blur": (1, 1, (1, 1, 1, 1), (64, 513, 513, 1)), "down2": (1, 2, (1, 1, 1, 1), (64, 513, 513, 1)), "up2": (2, 1, (2, 1, 2, 1), (12, 512, 512, 1))}[case]
xs = [torch.randn(shape, device=dev) for _ in range(10)]
for mode, seg in ((0, 0), (1, 0)):
    L.hav_lab_upfirdn2d(mode, seg)
    for x in xs:
        y = upfirdn2d.upfirdn2d(x, k4, up, up, dn, dn, *pad)
    torch.cuda.synchronize()
n = (xs[0].numel() + y.numel()) // 2
zs = [torch.randn(n, device=dev) for _ in range(10)]; e = zs[0].new_empty(0)
for z in zs:
    fused.fused_bias_act(z, e, e, 3, 0, 0.2, 1.0)
torch.cuda.synchronize()
