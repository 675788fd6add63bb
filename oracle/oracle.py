"""ctypes/numpy front end of oracle/_build/libhav_oracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under havatar_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhav_oracle.so")


class HavRenderParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("R", C.c_int32), ("ray_stride", C.c_int32), ("S_c", C.c_int32),
                ("S_f", C.c_int32), ("perturb", C.c_int32), ("noise_std", C.c_float),
                ("plane_res", C.c_int32), ("plane_ch", C.c_int32), ("vol_res", C.c_int32),
                ("nerf_scale", C.c_float * 3), ("nerf_trans", C.c_float * 3),
                ("skin_scale", C.c_float * 3), ("skin_trans", C.c_float * 3),
                ("seed", C.c_uint64), ("rng_offset", C.c_uint64), ("mlp_mode", C.c_int32), ("flags", C.c_int32),
                ("rng_counter", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64),
                ("dbg_zfine", C.c_void_p), ("status", C.c_void_p), ("grid_blocks", C.c_int32), ("reserved0", C.c_int32)]


class HavMlpWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")]


class OrcDebug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("z_coarse", "w_coarse", "z_fine", "w_fine", "raw_coarse", "raw_fine")]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("hav_oracle.c", "hav_oracle_impl.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_now.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def make_params(B, R, S_c, S_f, perturb, noise_std, nerf_scale, nerf_trans, skin_scale, skin_trans,
                ray_stride=8, plane_res=128, plane_ch=64, vol_res=64, seed=0, rng_offset=0):
    p = HavRenderParams()
    p.B, p.R, p.ray_stride, p.S_c, p.S_f = B, R, ray_stride, S_c, S_f
    p.perturb, p.noise_std = int(bool(perturb)), float(noise_std)
    p.plane_res, p.plane_ch, p.vol_res = plane_res, plane_ch, vol_res
    for i in range(3):
        p.nerf_scale[i], p.nerf_trans[i] = nerf_scale[i], nerf_trans[i]
        p.skin_scale[i], p.skin_trans[i] = skin_scale[i], skin_trans[i]
    p.seed, p.rng_offset = seed, rng_offset
    return p


def render_rays(scene, S_c, S_f, perturb=False, noise_std=0.0, t_rand=None, u_rand=None, noise_c=None,
                noise_f=None, f64=False, nthreads=1, debug=False):
    """Oracle for Trainer.predict_and_render_radiance on a `havatar_amd.synth.scene`-style dict.

    Returns dict(rgb_coarse [B,R,67], depth_coarse, acc_coarse, weights_max, rgb_fine, depth_fine, acc_fine)
    (+ debug arrays) as float32 (float64 if f64)."""
    rays = _f32(scene["rays"])
    B, R, stride = rays.shape
    bg = _f32(scene.get("bg"))
    inv_T = _f32(scene["inv_T"])
    planes = _f32(scene["planes"])
    vol = _f32(scene["vol"])
    m = {k: _f32(v) for k, v in scene["mlp"].items()}
    w = HavMlpWeights(*[_p(m[k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
    p = make_params(B, R, S_c, S_f, perturb, noise_std, scene["nerf_scale"], scene["nerf_trans"],
                    scene["skin_scale"], scene["skin_trans"], ray_stride=stride,
                    plane_res=planes.shape[-1], plane_ch=planes.shape[2], vol_res=vol.shape[-1])
    dt = np.float64 if f64 else np.float32
    S_fp = ((S_c + 1) // 2 + S_f) if S_f > 0 else 0
    out = {"rgb_coarse": np.zeros((B, R, 67), dt), "depth_coarse": np.zeros((B, R), dt),
           "acc_coarse": np.zeros((B, R), dt), "weights_max": np.zeros((B, R), dt),
           "rgb_fine": np.zeros((B, R, 67), dt), "depth_fine": np.zeros((B, R), dt),
           "acc_fine": np.zeros((B, R), dt)}
    dbg = None
    if debug:
        dbg_arr = {"z_coarse": np.zeros((B * R, S_c), dt), "w_coarse": np.zeros((B * R, S_c), dt),
                   "z_fine": np.zeros((B * R, max(S_fp, 1)), dt), "w_fine": np.zeros((B * R, max(S_fp, 1)), dt),
                   "raw_coarse": np.zeros((B * R, S_c, 68), dt), "raw_fine": np.zeros((B * R, max(S_fp, 1), 68), dt)}
        dbg = OrcDebug(*[_p(dbg_arr[k]) for k in ("z_coarse", "w_coarse", "z_fine", "w_fine", "raw_coarse", "raw_fine")])
        out.update(dbg_arr)
    t_rand, u_rand, noise_c, noise_f = _f32(t_rand), _f32(u_rand), _f32(noise_c), _f32(noise_f)
    fn = lib().orc_render_rays_f64 if f64 else lib().orc_render_rays_f32
    rc = fn(C.byref(p), _p(rays), _p(bg), _p(inv_T), _p(planes), _p(vol), C.byref(w),
            _p(t_rand), _p(u_rand), _p(noise_c), _p(noise_f),
            _p(out["rgb_coarse"]), _p(out["depth_coarse"]), _p(out["acc_coarse"]), _p(out["weights_max"]),
            _p(out["rgb_fine"]), _p(out["depth_fine"]), _p(out["acc_fine"]),
            C.byref(dbg) if dbg is not None else None, C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"orc_render_rays failed: {rc}")
    if S_f == 0:
        for k in ("rgb_fine", "depth_fine", "acc_fine"):
            out[k] = None
    return out


def fused_bias_act(x, b, ref, act, grad, alpha, scale):
    """Oracle of fused.fused_bias_act on numpy arrays (float32 or float64), NCHW-like x, bias [C]."""
    dt = np.float64 if x.dtype == np.float64 else np.float32
    x = np.ascontiguousarray(x, dt)
    out = np.empty_like(x)
    b = None if b is None or b.size == 0 else np.ascontiguousarray(b, dt)
    ref = None if ref is None or ref.size == 0 else np.ascontiguousarray(ref, dt)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    size_b = 0 if b is None else b.size
    # the reference op takes `float alpha, float scale` and casts them to scalar_t inside the kernel
    # (model/op/fused_bias_act_kernel.cu:67-70,98-101): also for float64 inputs they carry float32 precision
    alpha, scale = float(np.float32(alpha)), float(np.float32(scale))
    if dt == np.float64:
        fn, ct = lib().orc_fused_bias_act_f64, C.c_double
    else:
        fn, ct = lib().orc_fused_bias_act_f32, C.c_float
    rc = fn(_p(out), _p(x), _p(b), _p(ref), C.c_int(act), C.c_int(grad), ct(alpha), ct(scale),
            C.c_int64(x.size), C.c_int64(step_b), C.c_int64(size_b))
    if rc != 0:
        raise RuntimeError(f"orc_fused_bias_act failed: {rc}")
    return out


def upfirdn2d(x, k, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    """Oracle of upfirdn2d.upfirdn2d on x [major,in_h,in_w,minor] (float32/float64), k [kh,kw]."""
    dt = np.float64 if x.dtype == np.float64 else np.float32
    x = np.ascontiguousarray(x, dt)
    k = np.ascontiguousarray(k, np.float32)
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
    out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
    out = np.empty((major, out_h, out_w, minor), dt)
    fn = lib().orc_upfirdn2d_f64 if dt == np.float64 else lib().orc_upfirdn2d_f32
    rc = fn(_p(out), _p(x), _p(k), C.c_int64(major), in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
            px0, px1, py0, py1)
    if rc != 0:
        raise RuntimeError(f"orc_upfirdn2d failed: {rc}")
    return out


def eval_sh(deg, sh, dirs):
    sh = np.ascontiguousarray(sh, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    n, Cc, K = sh.shape
    out = np.empty((n, Cc), np.float32)
    rc = lib().orc_eval_sh_f32(deg, _p(sh), _p(dirs), C.c_int64(n), Cc, _p(out))
    if rc != 0:
        raise RuntimeError(f"orc_eval_sh failed: {rc}")
    return out


def gen_rays(H, W, intr, c2w, near, far, y0=0, y1=None):
    y1 = H if y1 is None else y1
    out = np.empty(((y1 - y0) * W, 8), np.float32)
    intr = np.ascontiguousarray(intr, np.float32)
    c2w = np.ascontiguousarray(c2w, np.float32).reshape(12)
    rc = lib().orc_gen_rays_f32(_p(out), H, W, _p(intr), _p(c2w), C.c_float(near), C.c_float(far), y0, y1)
    if rc != 0:
        raise RuntimeError(f"orc_gen_rays failed: {rc}")
    return out


def resample_depths(z, weights, S_f, zeta=None, dtype=np.float32):
    """(z_samples [n,S_f], z2 [n, ceil(S_c/2)+S_f]) of model/nerf_trainer.py:166-170 + utils/nerf_util.py:76-117; zeta = the raw
    torch.rand draw [n,S_f] or None (det=True)."""
    z = np.ascontiguousarray(z, dtype)
    weights = np.ascontiguousarray(weights, dtype)
    n, S_c = z.shape
    zs = np.empty((n, S_f), dtype)
    z2 = np.empty((n, (S_c + 1) // 2 + S_f), dtype)
    zt = None if zeta is None else np.ascontiguousarray(zeta, np.float32)
    fn = lib().orc_resample_depths_f32 if dtype == np.float32 else lib().orc_resample_depths_f64
    rc = fn(_p(z), _p(weights), C.c_int64(n), S_c, S_f, _p(zt) if zt is not None else None, _p(zs), _p(z2))
    if rc != 0:
        raise RuntimeError(f"orc_resample_depths failed: {rc}")
    return zs, z2
