"""CPU: the Python surface of havatar_amd.model.op on CPU tensors (pure-PyTorch branches, as in the reference)
against vectors produced by the reference's own CPU implementations."""
import os

import numpy as np
import torch

from helpers import GOLDEN, linf


def test_fused_leaky_relu_cpu_matches_reference():
    from havatar_amd.model.op import FusedLeakyReLU, fused_leaky_relu
    g = np.load(os.path.join(GOLDEN, "ops_reference_cpu.npz"))
    x, b = torch.from_numpy(g["fba_x"]), torch.from_numpy(g["fba_b"])
    assert linf(fused_leaky_relu(x, b).numpy(), g["fba_y"]) <= 1e-6
    assert linf(fused_leaky_relu(x).numpy(), g["fba_y_nobias"]) <= 1e-6
    # the CPU branch ignores negative_slope (reference fused_act.py:113): same numbers for any slope
    assert torch.equal(fused_leaky_relu(x, b, 0.7), fused_leaky_relu(x, b, 0.2))
    x2, b2 = torch.from_numpy(g["fba_x2"]), torch.from_numpy(g["fba_b2"])
    assert linf(fused_leaky_relu(x2, b2).numpy(), g["fba_y2"]) <= 1e-6
    m = FusedLeakyReLU(5)
    with torch.no_grad():
        m.bias.copy_(b)
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    gx, gb = torch.autograd.grad(y, (xr, m.bias), torch.from_numpy(g["fba_go"]))
    assert linf(gx.numpy(), g["fba_gx"]) <= 1e-6 and linf(gb.numpy(), g["fba_gb"]) <= 1e-4


def test_upfirdn2d_cpu_matches_reference():
    from havatar_amd.model.op import upfirdn2d
    g = np.load(os.path.join(GOLDEN, "ops_reference_cpu.npz"))
    x = torch.from_numpy(g["ufd_x"])
    for n in sorted({k[4:-2] for k in g.files if k.startswith("ufd_") and k.endswith("_k")}):
        k = torch.from_numpy(g[f"ufd_{n}_k"])
        ux, uy, dx, dy, px0, px1, py0, py1 = [int(v) for v in g[f"ufd_{n}_args"]]
        xr = x.clone().requires_grad_(True)
        y = upfirdn2d(xr, k, up=(ux, uy), down=(dx, dy), pad=(px0, px1, py0, py1))
        assert linf(y.detach().numpy(), g[f"ufd_{n}_y"]) <= 2e-6, n
        gi, = torch.autograd.grad(y, xr, torch.from_numpy(g[f"ufd_{n}_go"]))
        assert linf(gi.numpy(), g[f"ufd_{n}_gx"]) <= 5e-6, n
    # int up/down and 2-tuple pad are expanded like the reference (upfirdn2d.py:154-161)
    k = torch.from_numpy(g["ufd_m1_blur_pad22_k"])
    assert torch.equal(upfirdn2d(x, k, up=1, down=1, pad=(2, 2)), upfirdn2d(x, k, (1, 1), (1, 1), (2, 2, 2, 2)))
