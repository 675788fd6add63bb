"""Importance resampling between the two passes of the training forward (model/nerf_trainer.py:166-170 + utils/nerf_util.py:76-117):
the oracle against outputs of the reference itself (tests/golden/resample.npz, oracle/gen_golden_resample.py), and the HIP kernel
(hav_resample_depths, through the C ABI) bit-exact against the oracle."""
import numpy as np
import pytest

from oracle import oracle

G = None


def _golden():
    global G
    if G is None:
        import os
        G = np.load(os.path.join(os.path.dirname(__file__), "golden", "resample.npz"))
    return G


def _cases():
    return ["c64f64_rand", "c64f64_det", "c64f16_rand", "c33f7_rand", "c8f5_det", "c3f1_det"]


def _load(name):
    g = _golden()
    S_c, S_f, det, n = (int(v) for v in g[name + "/meta"])
    return g, S_c, S_f, bool(det), n, g[name + "/z"], g[name + "/w"], (None if det else g[name + "/zeta"])


def _conditioning(z, w, S_f, zeta):
    """per sample, from an fp64 statement of sample_pdf alone: the denominator it divides by (after the floor), the width of the bin
    it interpolates in, and whether the denominator sits ON the 1e-5 floor (|den - 1e-5| <= 1e-7: which side it falls on is decided
    by the last bit of a 62-term cumulative sum -- utils/nerf_util.py:112-113, SURVEY B-11; the reference's own fp32 is unstable there)"""
    z, w = z.astype(np.float64), w.astype(np.float64)[:, 1:-1] + 1e-5
    cdf = np.concatenate([np.zeros((w.shape[0], 1)), np.cumsum(w / w.sum(-1, keepdims=True), -1)], -1)
    if zeta is None:
        u = np.broadcast_to(np.linspace(0.0, 1.0, S_f)[None, :], (w.shape[0], S_f))
    else:
        u = np.arange(S_f)[None, :] * (1.0 / S_f) + zeta.astype(np.float64) * (1.0 / S_f - 1e-6)
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)
    below, above = np.maximum(inds - 1, 0), np.minimum(inds, cdf.shape[1] - 1)
    den = np.take_along_axis(cdf, above, 1) - np.take_along_axis(cdf, below, 1)
    mid = 0.5 * (z[:, 1:] + z[:, :-1])
    width = np.abs(np.take_along_axis(mid, above, 1) - np.take_along_axis(mid, below, 1))
    on_floor = np.abs(den - 1e-5) <= 1e-7
    # a knot of the CDF within the rounding of u / the cumulative sum: searchsorted may pick the neighbouring bin (continuous across the
    # knot except where one side is on the floor)
    return np.where(den < 1e-5, 1.0, den), width, on_floor, float(np.abs(np.diff(z, axis=-1)).max())


@pytest.mark.parametrize("name", _cases())
def test_oracle_fp64_equals_the_reference_fp64(name):
    g, S_c, S_f, det, n, z, w, zeta = _load(name)
    zs, z2 = oracle.resample_depths(z, w, S_f, zeta, dtype=np.float64)
    ref_zs, ref_z2 = g[name + "/zs_f64"], g[name + "/z2_f64"]
    err = np.abs(zs - ref_zs)
    body = err[:, :-1] if det else err
    assert body.size == 0 or body.max() <= 1e-9, body.max()
    if det:
        # u = 1 exactly: the CDF ends at 1 +- the rounding of its sum, so the last sample interpolates in the last bin or sits on its
        # end -- and where that bin is on the floor the two differ by one bin (tests/helpers.py::pdf_floor_sensitive says the same of
        # the march).  Held to one bin width; counted.
        bw = float(np.abs(np.diff(z.astype(np.float64), axis=-1)).max())
        assert err[:, -1].max() <= bw * 1.0001
        print(f"{name}: last deterministic sample off by a bin on {(err[:, -1] > 1e-9).sum()} of {n} rays")
    else:
        assert np.abs(z2 - ref_z2).max() <= 1e-9
    assert (np.diff(z2, axis=-1) >= 0).all() and z2.shape == (n, (S_c + 1) // 2 + S_f)


@pytest.mark.parametrize("name", _cases())
def test_oracle_fp32_against_the_reference_fp32(name):
    """fp32: ATen's sum is a vectorised cascade, the oracle's runs in index order -- the CDFs differ in their last bits (<= 62 * 2^-24)
    and the interpolation divides by the CDF increment.  Bar per sample: 4 ulp of the depth + bin width x min(1, 8e-6 / den); samples
    whose denominator is ON the 1e-5 floor: one bin (counted, and the reference's own fp32-vs-fp64 error is printed beside)."""
    g, S_c, S_f, det, n, z, w, zeta = _load(name)
    zs, z2 = oracle.resample_depths(z, w, S_f, zeta)
    ref, ref64 = g[name + "/zs_f32"], g[name + "/zs_f64"]
    den, width, on_floor, bw = _conditioning(z, w, S_f, zeta)
    if det:
        on_floor[:, -1] = True
    err = np.abs(zs.astype(np.float64) - ref)
    tol = 4 * np.spacing(np.float32(np.abs(z).max())) + width * np.minimum(1.0, 8e-6 / den)
    ok = (err <= tol) | on_floor
    assert ok.all(), (err[~ok].max(), np.argwhere(~ok)[:5])
    assert (err[on_floor] <= bw * 1.0001).all()
    exact = float((err == 0).mean())
    assert exact >= 0.7, exact
    print(f"{name}: {exact:.3f} of the samples bit-identical to the reference's fp32; {int(on_floor.sum())} of {err.size} on the floor "
          f"(worst {err[on_floor].max() if on_floor.any() else 0.0:.2e}; reference fp32 vs its fp64 there: "
          f"{np.abs(ref - ref64)[on_floor].max() if on_floor.any() else 0.0:.2e}); elsewhere worst {err[~on_floor].max() if (~on_floor).any() else 0.0:.2e}")


def test_oracle_rejects_bad_sizes():
    z = np.zeros((2, 2), np.float32)
    with pytest.raises(RuntimeError):
        oracle.resample_depths(z, z, 4, None)


@pytest.mark.parametrize("jitter", [False, True])
def test_oracle_resampling_equals_the_march_oracles_own(jitter):
    """orc_resample_depths on the coarse depths / weights the march oracle (orc_render_rays, pinned to the reference's whole-frame outputs)
    reports must give that oracle's merged fine depths: the two restate the same reference lines; they differ only in FMA contraction
    (the march oracle is compiled with it, this function without: one ulp of the depth in fp32, 4e-15 in fp64)."""
    from havatar_amd import synth
    sc = synth.scene(16, 16, "primary")
    n = sc["rays"].shape[1]
    okw = {"t_rand": synth.uniform((1, n, 64), 11), "u_rand": synth.uniform((n, 16), 12)} if jitter else {}
    for f64, tol in ((True, 4e-15), (False, 4.8e-7)):
        d = oracle.render_rays(sc, 64, 16, perturb=jitter, nthreads=4, f64=f64, debug=True, **okw)
        _, z2 = oracle.resample_depths(d["z_coarse"], d["w_coarse"], 16, okw.get("u_rand"), dtype=np.float64 if f64 else np.float32)
        err = np.abs(z2.astype(np.float64) - d["z_fine"])
        assert err.max() <= tol, (f64, err.max())
        assert (err == 0).mean() >= 0.99


def test_the_training_wrappers_refuse_cpu_tensors_instead_of_falling_back():
    """native/train_ops.py::resample_depths / equal_linear are HIP-only like the reference's CHECK_INPUT: a CPU tensor raises (the CPU
    statement of the same steps lives in the Trainer / EqualLinear modules, never behind these entry points)."""
    import torch
    from havatar_amd.native.train_ops import equal_linear, equal_linear_eligible, resample_depths
    z = torch.zeros(4, 64)
    with pytest.raises(RuntimeError, match="HIP float32 tensors only"):
        resample_depths(z, z, 16, None)
    x, W, b = torch.zeros(2, 32, requires_grad=True), torch.zeros(64, 32, requires_grad=True), torch.zeros(64, requires_grad=True)
    assert not equal_linear_eligible(x, W, b)
    with pytest.raises(RuntimeError, match="HIP float32 tensors only"):
        equal_linear(x, W, b, 1.0, 1.0)
    # the module on CPU tensors: the reference's statement, unchanged
    from havatar_amd.model.styleUnet import EqualLinear
    m = EqualLinear(32, 64)
    y = m(x)
    assert type(y.grad_fn).__name__ == "AddmmBackward0"
    assert torch.equal(y, torch.nn.functional.linear(x, m.weight * m.scale, m.bias * m.lr_mul))


# ------------------------------------------------------------------------------------------------ GPU
def _hip(z, w, S_f, zeta, samples=True):
    import torch
    from havatar_amd.native.train_ops import resample_depths
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    z2, zs = resample_depths(t(z), t(w), S_f, t(zeta), return_samples=True)
    torch.cuda.synchronize()
    return zs.cpu().numpy(), z2.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", _cases())
def test_hip_resample_bit_exact_against_the_oracle_on_the_reference_vectors(name):
    g, S_c, S_f, det, n, z, w, zeta = _load(name)
    zs, z2 = _hip(z, w, S_f, zeta)
    ozs, oz2 = oracle.resample_depths(z, w, S_f, zeta)
    assert np.array_equal(zs, ozs) and np.array_equal(z2, oz2)
    # and against the reference's own fp32 output under the bar the oracle is held to
    den, width, on_floor, bw = _conditioning(z, w, S_f, zeta)
    if det:
        on_floor[:, -1] = True
    err = np.abs(zs.astype(np.float64) - g[name + "/zs_f32"])
    tol = 4 * np.spacing(np.float32(np.abs(z).max())) + width * np.minimum(1.0, 8e-6 / den)
    assert ((err <= tol) | on_floor).all() and (err[on_floor] <= bw * 1.0001).all()


@pytest.mark.gpu
@pytest.mark.parametrize("S_c,S_f,det,n", [(64, 64, False, 4096), (64, 64, True, 4096), (64, 16, False, 10000), (128, 64, False, 777),
                                           (5, 125, False, 130), (3, 1, False, 5)])
def test_hip_resample_bit_exact_against_the_oracle_at_training_sizes(S_c, S_f, det, n):
    """config 5's size (4 096 rays x 64 + 64) and the limits of the kernel (S_c = 128, 128 merged depths, ragged tail blocks);
    ties: a ray whose weights are all zero and a ray whose depths repeat"""
    rng = np.random.default_rng(S_c * 1000 + S_f)
    z = np.sort(rng.uniform(3.5, 5.5, (n, S_c)).astype(np.float32), axis=-1)
    w = (rng.uniform(0, 1, (n, S_c)) ** 8 * (rng.uniform(0, 1, (n, S_c)) < 0.3)).astype(np.float32)
    w[0] = 0.0
    z[1] = z[1, 0]
    w[2, :] = 0.0
    w[2, S_c // 2] = 1.0
    zeta = None if det else rng.uniform(0, 1, (n, S_f)).astype(np.float32)
    zs, z2 = _hip(z, w, S_f, zeta)
    ozs, oz2 = oracle.resample_depths(z, w, S_f, zeta)
    assert np.array_equal(zs, ozs)
    assert np.array_equal(z2, oz2)
    assert (np.diff(z2, axis=-1) >= 0).all()
    # the merged list is a permutation of the even coarse depths and the new samples
    assert np.array_equal(np.sort(np.concatenate([z[:, ::2], zs], -1), -1), z2)


@pytest.mark.gpu
def test_hip_resample_refuses_what_it_cannot_hold_and_takes_empty_input():
    import torch
    from havatar_amd.native.train_ops import resample_depths
    dev = torch.device("cuda:0")
    z = torch.zeros(4, 130, device=dev)
    with pytest.raises(RuntimeError, match="HAV_EUNSUP"):
        resample_depths(z, z, 8, None)
    e = torch.zeros(0, 64, device=dev)
    assert resample_depths(e, e, 16, None).shape == (0, 48)
    with pytest.raises(RuntimeError):
        resample_depths(torch.zeros(4, 64), torch.zeros(4, 64), 16, None)        # CPU tensors: refused like the reference's CHECK_INPUT


@pytest.mark.gpu
def test_hip_resample_writes_every_slot_when_the_inputs_hold_nans():
    """a diverged step hands NaN weights / depths to the resampling.  The merge is a rank sort; its order is total (NaN last, index
    as tie-break -- torch.sort's order), so every slot of the merged list is written exactly once: the output (allocated with torch.empty)
    is a permutation of the even coarse depths and the new samples, finite values ascending, NaNs behind them -- what
    torch.sort(torch.cat(...)) of the replaced ATen statement returns."""
    import torch
    from havatar_amd.native.train_ops import resample_depths
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n, S_c, S_f = 64, 64, 16
    z = np.sort(rng.uniform(3.5, 5.5, (n, S_c)).astype(np.float32), axis=-1)
    w = rng.uniform(0, 1, (n, S_c)).astype(np.float32)
    w[0, 7] = np.nan              # NaN weight: the pdf, the CDF and every new sample of the ray are NaN
    z[1, 10] = np.nan             # NaN depths among the even coarse samples (and in two bin centres)
    z[1, 20] = np.nan
    w[2, :] = np.nan
    z[2, ::2] = np.nan            # everything NaN
    zt, wt = torch.from_numpy(z).to(dev), torch.from_numpy(w).to(dev)
    for _ in range(3):            # fresh torch.empty outputs over poisoned memory: a slot left unwritten would show the poison
        poison = torch.full((n, S_c // 2 + S_f), -12345.0, device=dev)
        del poison
        z2, zs = resample_depths(zt, wt, S_f, None, return_samples=True)
        torch.cuda.synchronize()
        z2, zs = z2.cpu().numpy(), zs.cpu().numpy()
        assert not (z2 == -12345.0).any()
        want = np.sort(np.concatenate([z[:, ::2], zs], -1), -1)          # numpy sorts NaN last, like torch.sort
        assert np.array_equal(z2, want, equal_nan=True)
        assert np.isnan(z2[0]).sum() == S_f and np.isnan(z2[2]).all() and np.isnan(z2[1]).sum() >= 2
        assert np.isfinite(z2[3:]).all()
