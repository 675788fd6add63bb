"""Shared helpers for the parity tests (oracle side + HIP side)."""
import os

import numpy as np

from havatar_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_KEYS = ("rgb_coarse", "depth_coarse", "acc_coarse", "weights_max", "rgb_fine", "depth_fine", "acc_fine")


def checksum(a):
    a = np.asarray(a, np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * np.arange(1, a.size + 1) % 7.0).sum()])


def load_render_fixture(name):
    """-> (fixture npz, scene dict with regenerated planes/vol/mlp, kwargs for the random inputs)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    H, W, recipe = int(g["meta_H"]), int(g["meta_W"]), str(g["meta_recipe"])
    B = g["rays"].shape[0]
    sc = synth.scene(H, W, recipe, B=B)
    sc["rays"], sc["bg"], sc["inv_T"] = g["rays"], g["bg"], g["inv_T"]
    # the regenerated seed tensors must be the ones the reference was run on
    np.testing.assert_allclose(checksum(sc["planes"]), g["cks_planes"], rtol=1e-12)
    np.testing.assert_allclose(checksum(sc["vol"]), g["cks_vol"], rtol=1e-12)
    ck = np.stack([checksum(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
    np.testing.assert_allclose(ck, g["cks_mlp"], rtol=1e-12)
    kw = {k: g[k] for k in ("t_rand", "u_rand", "noise_c", "noise_f") if k in g.files}
    cfg = dict(S_c=int(g["S_c"]), S_f=int(g["S_f"]), perturb=bool(g["perturb"]), noise_std=float(g["noise_std"]))
    return g, sc, cfg, kw


def hip_render(sc, S_c, S_f, perturb=False, noise_std=0.0, t_rand=None, u_rand=None, noise_c=None, noise_f=None,
               dbg_zfine=False, mlp="half", coarse_outputs=True, flags=None):
    """Run the HIP ray march (through the C ABI) on a synth-style scene; returns numpy outputs keyed like OUT_KEYS, plus
    "variant" (the kernel instantiation that was launched) and "fp16_fallback" (did the fp16 range guard hand over to bf16)."""
    import torch
    from havatar_amd.render import RayMarcher
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    from havatar_amd import _lib
    rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
    from havatar_amd.render import MLP_MODES
    rm.mlp_mode = MLP_MODES[mlp]
    if flags is not None:
        rm.flags = flags
    m = sc["mlp"]
    rm.set_mlp(*[t(m[k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
    rm.set_triplane(t(sc["planes"]))
    res = rm.render(t(sc["rays"]), t(sc["bg"]), t(sc["inv_T"]), t(sc["vol"]), S_c, S_f, perturb=perturb,
                    noise_std=noise_std, t_rand=t(t_rand), u_rand=t(u_rand), noise_c=t(noise_c), noise_f=t(noise_f),
                    dbg_zfine=dbg_zfine, coarse_outputs=coarse_outputs)
    torch.cuda.synchronize()
    out = {k: (None if v is None else v.cpu().numpy().reshape(v.shape[0], v.shape[1], -1)) for k, v in zip(OUT_KEYS, res)}
    if dbg_zfine:
        out["z_fine"] = res[7].cpu().numpy()
    out["variant"] = rm.last_variant
    out["fp16_fallback"] = rm.fp16_fallback_happened()
    return out


def linf(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64).reshape(np.shape(a))).max())


def pdf_floor_sensitive(sub, jitter, okw, nth=8):
    """bool [n]: rays whose inverse-CDF resampling sits ON sample_pdf's one true discontinuity (or, with deterministic depths, on its end
    point: below) -- `denom[denom < 1e-5] = 1`
    (utils/nerf_util.py:112-113).  A bin of an empty stretch holds (0 + 1e-5) / sum(w + 1e-5) of the CDF: on a ray whose coarse opacity is
    ~1 that increment IS 1e-5, and whether it compares below the floor is decided by the last bit of a cumulative sum -- the importance
    sample then lands at the bin's start (denominator replaced by 1) or anywhere inside it.  No fp32 evaluation of the path is stable there,
    the reference's included; which rays are affected follows from the fp64 oracle's coarse weights alone (independent of the kernel under
    test): a ray is flagged when a denominator it uses lies within 1e-7 of the floor (fp32 CDF values near 1 carry 6e-8 of rounding)."""
    from oracle import oracle
    d = oracle.render_rays(sub, 64, 16, perturb=jitter, nthreads=nth, f64=True, debug=True, **okw)
    w = d["w_coarse"][:, 1:-1] + 1e-5                                            # [n, 62]
    cdf = np.concatenate([np.zeros((w.shape[0], 1)), np.cumsum(w / w.sum(-1, keepdims=True), -1)], -1)      # [n, 63]
    ns = 16
    if jitter:
        u = np.arange(ns)[None, :] * (1.0 / ns) + okw["u_rand"].astype(np.float64) * (1.0 / ns - 1e-6)
    else:
        u = np.broadcast_to(np.linspace(0.0, 1.0, ns)[None, :], (w.shape[0], ns))
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1)                            # searchsorted(right=True)
    below, above = np.maximum(inds - 1, 0), np.minimum(inds, cdf.shape[1] - 1)
    den = np.take_along_axis(cdf, above, 1) - np.take_along_axis(cdf, below, 1)
    flagged = (np.abs(den - 1e-5) <= 1e-7).any(-1)
    slack = np.zeros(w.shape[0])
    if not jitter:
        # Deterministic sampling puts its last sample at u = 1 exactly: where the CDF ends at 1 -- give or take the rounding of a 62-term
        # fp32 cumulative sum (<= 62 * 2^-24 = 3.7e-6).  In exact arithmetic that sample is the last bin's centre; with cdf[-1] = 1 + eps
        # it is bins[-2] + (1 - eps / den) (bins[-1] - bins[-2]), den = the last bin's CDF increment (6e-5 when the bin is empty): the
        # sample moves by eps / den of the distance to its neighbour -- and dists[-1] repeats dists[-2] (utils/nerf_util.py:36-37), so
        # the opacities of the last TWO samples move by that fraction.  The slack a ray gets is what that does to its outputs: the fp64
        # oracle's weight of those two samples x min(1, 3.7e-6 / den) -- zero on every ray that is empty at the far end.
        den_last = cdf[:, -1] - cdf[:, -2]
        slack = d["w_fine"][:, -2:].sum(-1) * np.minimum(1.0, 3.7e-6 / np.maximum(den_last, 1e-12))
    return flagged, slack


def mlp_layer_errors(n_random=4096, seed=0):
    """{(layer, input kind, mode): (max, rms)} of |y - y64| / (sum_k |w_k x_k| + |b|): one dense layer of the radiance MLP evaluated by each
    arithmetic mode's own matrix routine (hav_debug_mlp_layer) against the fp64 product of the same fp32 weights and inputs.
    Input kinds: "random" -- dense inputs (relu of normals for layer 2, uniform [-1,1] for the positional-encoding columns of layer 1) and
    weights of mixed magnitude: the fp32 ACCUMULATION error of a 48- / 128-term sum dominates and every mode should look alike;
    "onehot" -- one non-zero input per query and zero biases, every k position, full-mantissa operands of magnitude [1/8, 2): the result is
    ONE product, so what is left is the mode's product error itself (fp32: 2^-24 from the final rounding; fp16 x 2: the dropped lo.lo /
    hi.tail / tail.hi terms, up to ~2^-22); "onehot_small" -- the same with weights in [2^-8, 2^-3): their fp16 lo parts are fp16
    SUBNORMALS (absolute operand error up to 2^-25 in the fp16 x 2 mode, include/havatar.h), which the MX mode's exact tails repair."""
    import torch
    from havatar_amd.render import MLP_MODES, RayMarcher
    from havatar_amd import synth
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    sc = synth.scene(8, 8, "primary")
    m = {k: np.array(v, np.float32) for k, v in sc["mlp"].items()}
    out = {}
    for kind in ("random", "onehot", "onehot_small"):
        mm = dict(m)
        for name in ("W1", "W2"):
            shp = m[name].shape
            if kind == "random":          # full-mantissa weights of mixed magnitude
                mm[name] = (rng.standard_normal(shp) * np.exp2(rng.integers(-6, 1, shp))).astype(np.float32)
            else:
                lo, hi = (-2, 2) if kind == "onehot" else (-7, -2)
                mm[name] = (rng.choice([-1.0, 1.0], shp) * rng.uniform(0.5, 1.0, shp) * np.exp2(rng.integers(lo, hi, shp))).astype(np.float32)
        if kind != "random":
            mm["b1"], mm["b2"] = np.zeros_like(m["b1"]), np.zeros_like(m["b2"])
        rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        rm.set_mlp(*[t(mm[k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
        for layer, K in ((1, 48), (2, 128)):
            W = mm["W1"][:, 128:176] if layer == 1 else mm["W2"]
            b = mm["b1"] if layer == 1 else mm["b2"]
            if kind == "random":
                x = rng.uniform(-1, 1, (n_random, K)) if layer == 1 else np.maximum(rng.standard_normal((n_random, K)), 0) * np.exp2(rng.integers(-3, 3, (n_random, 1)))
            else:
                reps = 8
                x = np.zeros((K * reps, K))
                vals = rng.uniform(0.5, 1.0, K * reps) * (np.exp2(rng.integers(-2, 2, K * reps)) if layer == 2 else np.exp2(rng.integers(-1, 1, K * reps)))
                if layer == 1:
                    vals *= rng.choice([-1.0, 1.0], K * reps)
                x[np.arange(K * reps), np.repeat(np.arange(K), reps)] = vals
            x = x.astype(np.float32)
            y64 = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
            mag = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64).T + np.abs(b).astype(np.float64)
            for name in ("f32", "split", "half", "mx"):
                y = rm.mlp_layer(t(x), layer, MLP_MODES[name]).cpu().numpy().astype(np.float64)
                e = np.abs(y - y64) / np.maximum(mag, 1e-30)
                out[(layer, kind, name)] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
    return out


def report(line):
    """Numbers a test measured on the way to its verdict (how many rays were in a loose class, their worst error): printed, and appended to
    gpurun_out/test_reports.txt so that a run on the GPU box brings them back (gpurun_out/ is scratch; the round's copy is profiles/r05_test_reports.txt)."""
    import os
    print(line)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_reports.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
