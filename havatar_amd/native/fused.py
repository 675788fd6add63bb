"""`fused` -- same call surface as the reference's pybind module (model/op/fused_bias_act.cpp:18-31),
backed by hav_fused_bias_act in libhavatar_hip.so.  HIP tensors only; no CPU path (the reference's module has none
either: CPU tensors never reach it, model/op/fused_act.py:108-119)."""
import ctypes as C

import torch

from .. import _lib

_DT = {torch.float32: _lib.HAV_F32, torch.float16: _lib.HAV_F16, torch.bfloat16: _lib.HAV_BF16,
       torch.float64: _lib.HAV_F64}


def _check_input(t, name):
    # CHECK_CUDA / CHECK_CONTIGUOUS (fused_bias_act.cpp:10-16) -> RuntimeError like TORCH_CHECK
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """out = scale * act'(input + bias[c]); `bias` / `refer` may be empty tensors (= absent)."""
    _check_input(input, "input")
    _check_input(bias, "bias")
    if input.dtype not in _DT:
        raise RuntimeError(f"fused_bias_act: unsupported dtype {input.dtype}")
    x = input
    b = bias.contiguous().to(x.dtype) if bias.numel() else None
    ref = refer.contiguous().to(x.dtype) if refer.numel() else None
    if ref is not None and ref.numel() != x.numel():
        raise RuntimeError("fused_bias_act: refer must have as many elements as input")
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d                                   # fused_bias_act_kernel.cu:82-88
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    with torch.cuda.device(x.device):                 # at::DeviceGuard (fused_bias_act.cpp:25)
        rc = _lib.lib().hav_fused_bias_act(
            C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()),
            C.c_void_p(b.data_ptr()) if b is not None else None,
            C.c_void_p(ref.data_ptr()) if ref is not None else None,
            _DT[x.dtype], int(act), int(grad), float(alpha), float(scale),
            x.numel(), step_b, b.numel() if b is not None else 0,
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_fused_bias_act")
    return out


def _f32c(t, name):
    _check_input(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t


def style_demod(style, mod_w, mod_b, wsq=None, eps=1e-8):
    """(s [B,Cin], d [B,Cout] | None): style vector and demodulation factors of one ModulatedConv2d in one launch
    (hav_style_demod; replaces EqualLinear + bias + square + matmul + eps + rsqrt of model/styleUnet.py:196-254)."""
    style, mod_w = _f32c(style, "style"), _f32c(mod_w, "mod_w")
    B, D = style.shape
    Cin = mod_w.shape[0]
    s = torch.empty(B, Cin, device=style.device, dtype=torch.float32)
    d = None
    Cout = 0
    if wsq is not None:
        wsq = _f32c(wsq, "wsq")
        Cout = wsq.shape[1]
        d = torch.empty(B, Cout, device=style.device, dtype=torch.float32)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(style.device):
        rc = _lib.lib().hav_style_demod(p(s), p(d), p(style), p(mod_w), p(_f32c(mod_b, "mod_b") if mod_b is not None else None), p(wsq),
                                        float(eps), B, D, Cin, Cout, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_style_demod")
    return s, d


class StylePlan:
    """Device-resident layer table of hav_style_demod_batched plus its output buffers: every modulated convolution of a generator
    in one launch.  entries: [(mod_w [Cin,D], mod_b [Cin] | None, wsq [Cin,Cout] | None, style_index)]; rebuilt by the caller when a
    weight-derived tensor moves (the key)."""

    def __init__(self, entries, B, device):
        import struct
        L = _lib.lib()
        self.key = self.make_key(entries, B, device)
        self.B, self.n = B, len(entries)
        self.keep = [t for e in entries for t in e[:3] if t is not None]          # the table holds raw pointers
        self.out, rows, first = [], [], 0
        self.max_cin = 0
        for mw, mb, wsq, idx in entries:
            mw = _f32c(mw, "mod_w")
            Cin = mw.shape[0]
            s = torch.empty(B, Cin, device=device, dtype=torch.float32)
            d = torch.empty(B, wsq.shape[1], device=device, dtype=torch.float32) if wsq is not None else None
            Cout = wsq.shape[1] if wsq is not None else 0
            ptr = lambda t: t.data_ptr() if t is not None else 0
            rows.append(struct.pack("<5Q4i", ptr(mw), ptr(mb), ptr(wsq), ptr(s), ptr(d), Cin, Cout, int(idx), first))
            first += L.hav_style_demod_blocks(Cout, 1 if wsq is not None else 0)
            self.max_cin = max(self.max_cin, Cin)
            self.out.append((s, d))
        self.total_blocks = first
        self.D = entries[0][0].shape[1]
        self.table = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(device)

    @staticmethod
    def make_key(entries, B, device):
        return (B, str(device)) + tuple((t.data_ptr() if t is not None else 0) for e in entries for t in e[:3]) + tuple(e[3] for e in entries)

    def run(self, styles, eps=1e-8):
        """styles [B, n_styles, D] -> [(s, d | None)] per layer (buffers owned by the plan, overwritten by the next run)."""
        styles = _f32c(styles, "styles")
        B, n_styles, D = styles.shape
        if B != self.B or D != self.D:
            raise RuntimeError("StylePlan.run: built for B=%d, D=%d" % (self.B, self.D))
        with torch.cuda.device(styles.device):
            rc = _lib.lib().hav_style_demod_batched(C.c_void_p(self.table.data_ptr()), self.n, self.total_blocks, self.max_cin,
                                                    C.c_void_p(styles.data_ptr()), float(eps), B, n_styles, D,
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "hav_style_demod_batched")
        return self.out


def styled_epilogue(x, demod, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    """leaky_relu((x * demod[b,c] + noise_weight * noise) + bias[c]) * scale in one pass (hav_styled_epilogue)."""
    x = _f32c(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    nb = 0
    if noise is not None:
        noise = _f32c(noise, "noise")
        if noise.numel() == B * H * W and B > 1:
            nb = 1
        elif noise.numel() != H * W:
            raise RuntimeError("styled_epilogue: noise must be [1,1,H,W] or [B,1,H,W]")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(x.device):
        rc = _lib.lib().hav_styled_epilogue(p(out), p(x), p(_f32c(demod, "demod") if demod is not None else None), p(noise),
                                            p(noise_weight), p(_f32c(bias, "bias") if bias is not None else None),
                                            float(negative_slope), float(scale), B, Cc, H * W, nb,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_styled_epilogue")
    return out


def torgb(x, weight, s, bias, skip, scale):
    """ToRGB's modulated 1x1 convolution + bias + skip add as one pass (hav_torgb; reference model/styleUnet.py:602-628).
    x [B,Cin,H,W], weight [Cout,Cin] (or [Cout,Cin,1,1]), s [B,Cin] | None, bias [Cout] (any shape with Cout elements) | None,
    skip [B,Cout,H,W] | None.  Returns None when the shape is not taken (the caller keeps its ATen route)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return None
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if Cout not in (3, 12) or Cin > 1024 or (H * W) % 4 or weight.numel() != Cout * Cin:
        return None
    # any layout the ATen route took is taken here too: strided / channels-last / sliced operands are made contiguous (a copy only when
    # needed); operands of another dtype or device, or not 16-byte aligned (the kernel's float4 loads), keep the caller's ATen route
    ok = lambda t: t is None or (t.is_cuda and t.dtype == torch.float32 and t.device == x.device)
    if not (ok(weight) and ok(s) and ok(bias) and ok(skip)):
        return None
    if skip is not None and tuple(skip.shape) != (B, Cout, H, W):
        raise RuntimeError("torgb: skip must be [B,Cout,H,W]")
    x = x.contiguous()
    w2 = weight.reshape(Cout, Cin).contiguous()
    s = s.contiguous() if s is not None else None
    bias = bias.reshape(-1).contiguous() if bias is not None else None
    skip = skip.contiguous() if skip is not None else None
    if any(t is not None and t.data_ptr() % 16 for t in (x, skip)):
        return None
    out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(x.device):
        rc = _lib.lib().hav_torgb(p(out), p(x), p(w2), p(s), p(bias), p(skip), float(scale), B, Cout, Cin, H * W,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc == -2:          # HAV_EUNSUP: a shape this build does not take -- not an error, the caller falls back
        return None
    _lib.check(rc, "hav_torgb")
    return out


def haar(x, k4, inverse=False):
    """HaarTransform / InverseHaarTransform of model/styleUnet.py as one launch (hav_haar_dwt / hav_haar_idwt), bit-identical to the
    four upfirdn2d calls.  k4 [4,2,2]: the kernels of the four calls.  Returns None when the shape is not taken (caller falls back)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return None
    x = x.contiguous()
    B, Cc, H, W = x.shape
    if inverse:
        if Cc % 4 or W % 4:
            return None
        out = torch.empty(B, Cc // 4, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        fn, Cn = _lib.lib().hav_haar_idwt, Cc // 4
    else:
        if H % 2 or W % 8:
            return None
        out = torch.empty(B, Cc * 4, H // 2, W // 2, device=x.device, dtype=torch.float32)
        fn, Cn = _lib.lib().hav_haar_dwt, Cc
    k4 = _f32c(k4, "k4")
    with torch.cuda.device(x.device):
        rc = fn(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(k4.data_ptr()), B, Cn, H, W,
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_haar")
    return out


def haar_up2(x, ki4, fir, kd4):
    """dwt(upsample(iwt(x))) of ToRGB's skip path (model/styleUnet.py:476-480) as one launch (hav_haar_up2): x [B,12,H,W] -> [B,12,2H,2W],
    bit-identical to haar(inverse) -> upfirdn2d(up 2, pad (2, 1)) -> haar.  ki4 / kd4 [4,2,2]: the synthesis / analysis kernels, fir [4,4]:
    Upsample's kernel.  Returns None when the shape is not taken (caller falls back)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(fir.shape) == (4, 4)):
        return None
    B, Cc, H, W = x.shape
    if Cc % 4 or W % 2:
        return None
    x = x.contiguous()
    out = torch.empty(B, Cc, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
    ki4, kd4, fir = _f32c(ki4, "ki4"), _f32c(kd4, "kd4"), _f32c(fir, "fir")
    with torch.cuda.device(x.device):
        rc = _lib.lib().hav_haar_up2(C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(ki4.data_ptr()), C.c_void_p(fir.data_ptr()),
                                     C.c_void_p(kd4.data_ptr()), B, Cc // 4, H, W, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_haar_up2")
    return out


def style_mlp(z, blob, n_layers, D, slope=0.2, gain=2 ** 0.5):
    """PixelNorm + n x (EqualLinear + fused leaky-ReLU) on z [B, D0 <= 64] in one launch (hav_style_mlp); blob: see include/havatar.h."""
    z = _f32c(z.contiguous(), "z")
    B, D0 = z.shape
    out = torch.empty(B, D, device=z.device, dtype=torch.float32)
    with torch.cuda.device(z.device):
        rc = _lib.lib().hav_style_mlp(C.c_void_p(out.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(blob.data_ptr()), int(n_layers), B, D0, int(D),
                                      float(slope), float(gain), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_style_mlp")
    return out
