"""Golden vectors for the importance resampling between the two passes (model/nerf_trainer.py:166-170 + utils/nerf_util.py:76-117).

Runs in the BUILD container only: imports the reference's `sample_pdf` from /root/reference and executes the three statements of
nerf_trainer.py:166-170 around it on CPU, float32 and float64.  torch.rand is re-seeded before every call so that the raw draw
(`zeta`) the reference makes inside sample_pdf is also stored.  Writes tests/golden/resample.npz (numbers only).

    python oracle/gen_golden_resample.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
from utils.nerf_util import sample_pdf  # noqa: E402  (the reference's)


def coarse_pass(n, S_c, seed, near=3.6, far=5.2, jitter=True):
    """depths like nerf_trainer.py:134-144 and compositing weights of a head-like density: empty space, one or two shells"""
    g = torch.Generator().manual_seed(seed)
    t = torch.linspace(0.0, 1.0, S_c, dtype=torch.float64)
    z = (near * (1.0 - t) + far * t).expand(n, S_c).clone()
    if jitter:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper, lower = torch.cat((mids, z[..., -1:]), -1), torch.cat((z[..., :1], mids), -1)
        z = lower + (upper - lower) * torch.rand(z.shape, generator=g, dtype=torch.float64)
    c = near + (far - near) * torch.rand(n, 1, generator=g, dtype=torch.float64)
    width = 0.02 + 0.3 * torch.rand(n, 1, generator=g, dtype=torch.float64)
    sigma = 40.0 * torch.exp(-0.5 * ((z - c) / width) ** 2) + (torch.rand(n, 1, generator=g, dtype=torch.float64) < 0.3) * 2.0 * torch.rand(n, S_c, generator=g, dtype=torch.float64)
    sigma[: n // 8] = 0.0                                  # rays through empty space: every CDF increment on the 1e-5 floor
    d = torch.cat((z[..., 1:] - z[..., :-1], z[..., -1:] - z[..., -2:-1]), -1)
    alpha = 1.0 - torch.exp(-sigma * d)
    T = torch.cumprod(torch.cat((torch.ones(n, 1, dtype=torch.float64), 1.0 - alpha + 1e-10), -1), -1)[:, :-1]
    return z, alpha * T


def run(z, w, S_f, det, seed):
    torch.manual_seed(seed)
    zeta = torch.rand(list(w.shape[:-1]) + [S_f], dtype=w.dtype)          # the draw sample_pdf is about to make (:95)
    torch.manual_seed(seed)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])                              # nerf_trainer.py:166
    z_s = sample_pdf(z_mid, w[..., 1:-1], S_f, det=det).detach()          # :167-168
    z2, _ = torch.sort(torch.cat((z[:, ::2], z_s), dim=-1), dim=-1)       # :170
    return zeta, z_s, z2


def main():
    out = {}
    cases = [("c64f64_rand", 64, 64, False, 256, 1), ("c64f64_det", 64, 64, True, 256, 2), ("c64f16_rand", 64, 16, False, 128, 3),
             ("c33f7_rand", 33, 7, False, 64, 4), ("c8f5_det", 8, 5, True, 32, 5), ("c3f1_det", 3, 1, True, 8, 6)]
    for name, S_c, S_f, det, n, seed in cases:
        z64, w64 = coarse_pass(n, S_c, seed, jitter=not det)
        z32, w32 = z64.float(), w64.float()
        zeta, zs32, z232 = run(z32, w32, S_f, det, 100 + seed)
        _, zs64, z264 = run(z32.double(), w32.double(), S_f, det, 100 + seed)       # fp64 statement on the SAME fp32 inputs
        if not det:     # the float64 draw differs from the float32 one: the fp64 statement is run on the fp32 draw (torch.rand patched)
            zs64, z264 = run_with_zeta(z32.double(), w32.double(), zeta)
        out[name + "/z"] = z32.numpy(); out[name + "/w"] = w32.numpy()
        out[name + "/zeta"] = zeta.numpy() if not det else np.zeros((0,), np.float32)
        out[name + "/zs_f32"] = zs32.numpy(); out[name + "/z2_f32"] = z232.numpy()
        out[name + "/zs_f64"] = zs64.numpy(); out[name + "/z2_f64"] = z264.numpy()
        out[name + "/meta"] = np.array([S_c, S_f, int(det), n], np.int64)
    path = os.path.join(ROOT, "tests", "golden", "resample.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def run_with_zeta(z, w, zeta):
    """the same three statements with torch.rand returning a given draw (cast to the dtype sample_pdf asks for)"""
    import unittest.mock as mock
    with mock.patch("torch.rand", lambda *a, **k: zeta.to(k.get("dtype", torch.float32))):
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        z_s = sample_pdf(z_mid, w[..., 1:-1], zeta.shape[-1], det=False).detach()
    z2, _ = torch.sort(torch.cat((z[:, ::2], z_s), dim=-1), dim=-1)
    return z_s, z2


if __name__ == "__main__":
    main()
