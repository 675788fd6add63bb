"""Torch-level host side of the fused ray march (one C-ABI call per set of rays).

`RayMarcher` owns the device-resident, kernel-friendly copies of the per-model / per-frame constants:
the MFMA-fragment-ordered weight blob (re-packed only when the weights change) and the prepared tri-planes
(once per frame: channels-last, pre-multiplied by the plane columns of layers_xyz.0), and launches
`hav_render_rays` on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

from . import _lib
from .graph import weights_epoch

_DEBUG_SYNC = os.environ.get("HAVATAR_DEBUG_SYNC", "0") == "1"


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_f32_cuda(name, t, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise RuntimeError(f"{name} is required")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor")     # TORCH_CHECK(is_cuda), fused_bias_act.cpp:10-16
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


# HAVATAR_MLP -> HavRenderParams.mlp_mode.  "mx": fp16 hi + lo for the three leading partial products + block-scaled 4- / 6-bit matrix
# instructions for the three terms of order 2^-22 (operands hi + lo + tail = the fp32 value; 0.65 of the bf16 split's matrix time);
# "split": every operand = hi + mid + lo bf16 exactly, six partial products; "half": the fp16 double split alone (22-bit operands:
# NARROWER than the reference's fp32, the fastest); "f32": exact fp32 MFMA (an fmaf chain).
MLP_MODES = {"mx": _lib.HAV_MLP_SPLIT_F16_MX, "split": _lib.HAV_MLP_SPLIT_BF16, "bf16": _lib.HAV_MLP_SPLIT_BF16, "half": _lib.HAV_MLP_SPLIT_F16,
             "fp16": _lib.HAV_MLP_SPLIT_F16, "f32": _lib.HAV_MLP_F32}
DEFAULT_MLP = "mx"


class RayMarcher:
    """Device state + launcher for `hav_render_rays`."""

    def __init__(self, nerf_scale, nerf_trans, skin_scale, skin_trans, plane_res=128, plane_ch=64, vol_res=64):
        self.nerf_scale, self.nerf_trans = tuple(map(float, nerf_scale)), tuple(map(float, nerf_trans))
        self.skin_scale, self.skin_trans = tuple(map(float, skin_scale)), tuple(map(float, skin_trans))
        self.plane_res, self.plane_ch, self.vol_res = plane_res, plane_ch, vol_res
        self.blob = None
        self._blob_key = None
        self.grid_blocks = 0          # > 0: the march runs on at most this many compute units (HavRenderParams.grid_blocks; tools/pipeline_probe.py)
        self.planes_cl = None
        self._planes_key = None
        self.rng_counter = None
        self.rng_offset = 0
        self.seed = 0x9E3779B97F4A7C15
        # arithmetic of the two dense layers (include/havatar.h): MLP_MODES / DEFAULT_MLP below
        self.mlp_mode = MLP_MODES.get(os.environ.get("HAVATAR_MLP", DEFAULT_MLP), MLP_MODES[DEFAULT_MLP])
        # fine-pass cache: re-use the coarse pass's field values for the even coarse samples the merged list repeats
        self.fine_cache = os.environ.get("HAVATAR_FINE_CACHE", "1") != "0"
        self._workspace = None
        # per-call switches of the library (HavRenderParams.flags).  The A/B environment variables are read HERE, once per
        # marcher; the library itself reads no environment on the launch path.
        self.flags = 0
        if os.environ.get("HAV_MARCH", "")[:1] == "p":
            self.flags |= _lib.HAV_FLAG_PAIR_KERNEL
        self.flags |= {"cache": _lib.HAV_FLAG_FINE_CACHE, "recompute": _lib.HAV_FLAG_FINE_RECOMPUTE}.get(os.environ.get("HAV_FINE", ""), 0)
        self._status = None          # device word the library ORs HAV_STATUS_* bits into
        self.last_variant = None     # name of the kernel instantiation the last render() launched

    # -- constants -----------------------------------------------------------------------------
    def set_mlp(self, W1, b1, W2, b2, Wa, ba, Wf, bf, Wc, bc, force=False):
        """Pack nn.Linear-layout weights (model/nerf_model.py:46-51) into the fragment-ordered blob."""
        ts = [_chk_f32_cuda(n, t) for n, t in zip(("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc"),
                                                   (W1, b1, W2, b2, Wa, ba, Wf, bf, Wc, bc))]
        key = tuple((t.data_ptr(), t._version) for t in ts) + (weights_epoch(),)
        if not force and key == self._blob_key and self.blob is not None:
            return
        L = _lib.lib()
        if self.blob is None or self.blob.device != ts[0].device:
            self.blob = torch.empty(int(L.hav_mlp_blob_bytes()) // 4, dtype=torch.float32, device=ts[0].device)
        w = _lib.HavMlpWeights(*[t.data_ptr() for t in ts])
        with torch.cuda.device(ts[0].device):
            _lib.check(L.hav_mlp_pack(_ptr(self.blob), C.byref(w), _stream()), "hav_mlp_pack")
        self._blob_key = key
        self._keep = ts

    def set_triplane(self, planes_nchw):
        """[2,B,C,H,W] (Trainer.model_coarse.triPlane_embeddings) -> prepared planes on the device (128 floats per texel of [2,B,H,W];
        the arrangement inside the buffer belongs to the library).

        Must be called after set_mlp() and again whenever the weights change (the projection uses them)."""
        if self.blob is None:
            raise RuntimeError("set_mlp() must be called before set_triplane()")
        p = _chk_f32_cuda("triPlane_embeddings", planes_nchw)
        two, B, Cc, H, W = p.shape
        if two != 2 or Cc != self.plane_ch or H != W:
            raise RuntimeError(f"unsupported tri-plane shape {tuple(p.shape)}")
        self.plane_res = H
        need = int(_lib.lib().hav_triplane_prepared_bytes(B, H, W)) // 4        # planes + the range-guard trailer
        if self.planes_cl is None or self.planes_cl.numel() != need or self.planes_cl.device != p.device:
            self.planes_cl = torch.empty(need, dtype=torch.float32, device=p.device)
        self._planes_B = B
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().hav_triplane_prepare(_ptr(self.planes_cl), _ptr(p), _ptr(self.blob), B, Cc, H, W, _stream()),
                       "hav_triplane_prepare")
        self._planes_key = self._blob_key

    # -- launch --------------------------------------------------------------------------------
    def render(self, rays, bg, inv_T, skin_vol, S_c, S_f, perturb=False, noise_std=0.0,
               t_rand=None, u_rand=None, noise_c=None, noise_f=None, dbg_zfine=False, coarse_outputs=True):
        """rays [B,R,>=8], bg [B,R,3]|None, inv_T [B,4,3], skin_vol [2,D,H,W] or [1,2,D,H,W].

        Returns the 7-tuple of predict_and_render_radiance (model/nerf_trainer.py:194-201):
        rgb_coarse [B,R,67], depth_coarse [B,R,1], acc_coarse [B,R,1], weights_max [B,R,1],
        rgb_fine, depth_fine, acc_fine (None x3 if S_f == 0).  coarse_outputs=False (with a fine pass): the caller does not need the
        coarse pass's composited maps; they come back as None where the kernel can skip them."""
        if self.blob is None or self.planes_cl is None:
            raise RuntimeError("set_mlp() and set_triplane() must be called before render()")
        if self._planes_key != self._blob_key:
            raise RuntimeError("weights changed since set_triplane(): call set_triplane() again")
        rays = _chk_f32_cuda("ray_batch", rays)
        B, R, stride = rays.shape
        bg = _chk_f32_cuda("background_prior", bg, allow_none=True)
        inv_T = _chk_f32_cuda("inv_head_T", inv_T)
        vol = _chk_f32_cuda("canonical_W", skin_vol)
        if vol.dim() == 5:
            vol = vol[0]
        if self._planes_B != B or inv_T.shape[0] != B:
            raise RuntimeError("batch mismatch between rays, inv_head_T and the tri-plane")
        dev = rays.device
        S_fp = (S_c + 1) // 2 + S_f if S_f > 0 else 0
        p = _lib.HavRenderParams()
        p.B, p.R, p.ray_stride, p.S_c, p.S_f = B, R, stride, S_c, S_f
        p.perturb, p.noise_std = int(bool(perturb)), float(noise_std)
        p.plane_res, p.plane_ch, p.vol_res = self.plane_res, self.plane_ch, vol.shape[-1]
        for i in range(3):
            p.nerf_scale[i], p.nerf_trans[i] = self.nerf_scale[i], self.nerf_trans[i]
            p.skin_scale[i], p.skin_trans[i] = self.skin_scale[i], self.skin_trans[i]
        p.seed, p.rng_offset = self.seed, self.rng_offset
        p.mlp_mode, p.flags = self.mlp_mode, self.flags
        p.grid_blocks = int(self.grid_blocks)
        if self._status is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        p.status = self._status.data_ptr()
        if self.rng_counter is None or self.rng_counter.device != dev:
            self.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)      # device-side call counter (graph-replay safe)
        p.rng_counter = self.rng_counter.data_ptr()
        # scratch for the fine-pass cache (the library never allocates): kept for the life of the marcher, grown on demand
        L = _lib.lib()
        need = int(L.hav_render_workspace_bytes(C.byref(p))) if self.fine_cache else 0
        if need > 0:
            if self._workspace is None or self._workspace.device != dev or self._workspace.numel() < need:
                try:
                    self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
                except torch.OutOfMemoryError:            # no room for the cache: evaluate every merged sample, like the reference
                    self._workspace, self.fine_cache, need = None, False, 0
            if need > 0:
                p.workspace, p.workspace_bytes = self._workspace.data_ptr(), self._workspace.numel()
        t_rand = _chk_f32_cuda("t_rand", t_rand, True)
        u_rand = _chk_f32_cuda("u_rand", u_rand, True)
        noise_c = _chk_f32_cuda("noise_c", noise_c, True)
        noise_f = _chk_f32_cuda("noise_f", noise_f, True)
        for nm, t, n in (("t_rand", t_rand, B * R * S_c), ("u_rand", u_rand, B * R * S_f),
                         ("noise_c", noise_c, B * R * S_c), ("noise_f", noise_f, B * R * S_fp)):
            if t is not None and t.numel() != n:
                raise RuntimeError(f"{nm} has {t.numel()} elements, expected {n}")
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        if not coarse_outputs and S_f > 0 and need > 0:          # declined (only the block kernel can skip them: need > 0 implies it)
            rgb_c = d_c = a_c = None
        else:
            rgb_c, d_c, a_c = e(B, R, 67), e(B, R, 1), e(B, R, 1)
        wmax = e(B, R, 1)
        if S_f > 0:
            rgb_f, d_f, a_f = e(B, R, 67), e(B, R, 1), e(B, R, 1)
        else:
            rgb_f = d_f = a_f = None
        out = _lib.HavRenderOut(*(t.data_ptr() if t is not None else None for t in (rgb_c, d_c, a_c, wmax, rgb_f, d_f, a_f)))
        zf = None
        with torch.cuda.device(dev):
            if dbg_zfine and S_fp > 0:
                zf = e(int(dbg_zfine) * B * R, S_fp)          # (an int > 1: room for the extra planes a -DHAV_DEBUG_DUMP3 build writes)
                p.dbg_zfine = zf.data_ptr()
            rc = L.hav_render_rays(C.byref(p), _ptr(rays), _ptr(bg), _ptr(inv_T), _ptr(self.planes_cl), _ptr(vol),
                                   _ptr(self.blob), _ptr(t_rand), _ptr(u_rand), _ptr(noise_c), _ptr(noise_f),
                                   C.byref(out), _stream())
            _lib.check(rc, "hav_render_rays")
            if _DEBUG_SYNC:
                torch.cuda.synchronize()
        self.last_variant = self._variant_of(p, rgb_c is not None, any(t is not None for t in (t_rand, u_rand, noise_c, noise_f)))
        res = (rgb_c, d_c, a_c, wmax, rgb_f, d_f, a_f)
        return res + (zf,) if dbg_zfine else res

    def mlp_layer(self, x, layer, mode=None):
        """Test hook (hav_debug_mlp_layer): y [n,128] = W . x + b of dense layer 1 (x [n,48]: the positional-encoding columns) or 2
        (x [n,128]) WITHOUT the activation, evaluated by the march kernel's matrix routine of `mode` (default: this marcher's)."""
        x = _chk_f32_cuda("x", x)
        n = x.shape[0]
        if x.dim() != 2 or x.shape[1] != (48 if layer == 1 else 128):
            raise RuntimeError("mlp_layer: x must be [n,48] (layer 1) or [n,128] (layer 2)")
        y = torch.empty(n, 128, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hav_debug_mlp_layer(_ptr(y), _ptr(x), _ptr(self.blob), self.mlp_mode if mode is None else mode, int(layer), n,
                                                      _stream()), "hav_debug_mlp_layer")
        return y

    def fp16_fallback_happened(self):
        """True if, since the last call of this method, the fp16 range guard made the bf16-split kernel render a call
        (HAV_STATUS_FP16_FALLBACK).  Synchronises: for tests and diagnostics, not for the frame loop."""
        if self._status is None:
            return False
        v = int(self._status.item())
        self._status.zero_()
        return bool(v & _lib.HAV_STATUS_FP16_FALLBACK)

    @staticmethod
    def _variant_of(p, coarse_outputs, injected):
        buf = C.create_string_buffer(64)
        _lib.check(_lib.lib().hav_render_variant_name(C.byref(p), int(bool(coarse_outputs)), int(bool(injected)), buf, 64),
                   "hav_render_variant_name")
        return buf.value.decode()

    def variant(self, S_c, S_f, perturb=False, noise_std=0.0, coarse_outputs=True, injected=False):
        """Kernel instantiation a render() with these settings launches (the library's own decision function)."""
        p = _lib.HavRenderParams()
        p.S_c, p.S_f, p.perturb, p.noise_std = S_c, S_f, int(bool(perturb)), float(noise_std)
        p.mlp_mode, p.flags = self.mlp_mode, self.flags
        p.B, p.R = 1, 1 << 18
        if self.fine_cache and S_f > 0:                  # same decision as render(): a workspace is always provided when useful
            p.workspace, p.workspace_bytes = 1, 1 << 62
        return self._variant_of(p, coarse_outputs, injected)


def gen_rays(H, W, intr, c2w, near, far, device, out=None):
    """[1, H*W, 8] rays (origin, unit direction, near, far) generated ON the device by hav_gen_rays: the device-side form of
    dataloader/data_util.py::get_rays + the near/far columns of dataloader/dataloader.py:174-181.  intr = (fx, fy, cx/W, cy/H),
    c2w = [3,4] camera-to-world.  Only 18 floats cross PCIe instead of the [H*W, 11] ray table."""
    if out is None:
        out = torch.empty(1, H * W, 8, device=device, dtype=torch.float32)
    i4 = (C.c_float * 4)(*[float(v) for v in intr])
    m12 = (C.c_float * 12)(*[float(v) for v in torch.as_tensor(c2w, dtype=torch.float32).reshape(-1)[:12]])
    with torch.cuda.device(out.device):
        rc = _lib.lib().hav_gen_rays(C.c_void_p(out.data_ptr()), int(H), int(W), i4, m12, float(near), float(far), 0, int(H), _stream())
    _lib.check(rc, "hav_gen_rays")
    return out
