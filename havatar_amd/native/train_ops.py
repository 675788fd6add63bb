"""Training-path field ops (hav_field_inputs_*, hav_composite_*): the HIP counterparts, under autograd, of what surrounds the
radiance MLP in predict_and_render_radiance -- skinning field + box warp + tri-plane gather + positional encoding on one side,
volume_render_radiance_field on the other.  HIP float32 tensors only; there is no fallback in here."""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

# Under torch.autocast these nodes compute in fp32 like everywhere else: inputs are cast to fp32 and autocast is off inside forward AND backward
# (without this, torch.mm inside Conv3dSmall.forward would return bf16 under autocast while backward, which runs outside it, multiplies a
# bf16 gradient with the saved fp32 patch matrix: dtype mismatch; the kernels themselves refuse anything but fp32).
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_hip(name, *ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(name + ": HIP float32 tensors only")


def _field_params(pts, planes_cl, vol, boxes):
    P, B, H, W, Cc = planes_cl.shape
    if P != 2 or pts.ndim != 3 or pts.shape[0] != B or pts.shape[-1] != 3:
        raise RuntimeError("field_inputs: planes [2,B,H,W,C], pts [B,N,3]")
    if vol.ndim != 5 or vol.shape[0] != 1 or vol.shape[1] != 2 or not (vol.shape[2] == vol.shape[3] == vol.shape[4]):
        raise RuntimeError("field_inputs: skinning volume [1,2,D,D,D]")
    p = _lib.HavFieldParams()
    p.n, p.n_per_b, p.B, p.H, p.W, p.C, p.D = B * pts.shape[1], pts.shape[1], B, H, W, Cc, vol.shape[2]
    for name, v in zip(("nerf_scale", "nerf_trans", "skin_scale", "skin_trans"), boxes):
        setattr(p, name, (C.c_float * 3)(*[float(x) for x in v]))
    return p


def deterministic():
    """HAVATAR_DETERMINISTIC=1: the training-side scatters sum in 64-bit fixed point (bit-reproducible gradients; ~1 ms per step)."""
    return os.environ.get("HAVATAR_DETERMINISTIC", "0") == "1"


class FieldInputs(Function):
    """X [B*N, 2C+48] = cat(triplane(boxwarp(p')), PE(p')),  p' = skinning_field(pts, inv_T, vol).  Gradients: planes_cl, vol."""

    @staticmethod
    @_fwd32
    def forward(ctx, pts, inv_T, vol, planes_cl, boxes, ray_rows=0):
        _need_hip("FieldInputs", pts, inv_T, vol, planes_cl)
        pts, inv_T, vol, planes_cl = pts.contiguous(), inv_T.contiguous(), vol.contiguous(), planes_cl.contiguous()
        p = _field_params(pts, planes_cl, vol, boxes)
        X = torch.empty(p.n, 2 * p.C + 48, device=pts.device, dtype=torch.float32)
        with torch.cuda.device(pts.device):
            rc = _lib.lib().hav_field_inputs_fwd(_p(X), C.byref(p), _p(pts), _p(inv_T), _p(vol), _p(planes_cl), _stream())
        _lib.check(rc, "hav_field_inputs_fwd")
        ctx.save_for_backward(pts, inv_T, vol, planes_cl)
        ctx.boxes, ctx.ray_rows = boxes, ray_rows
        return X

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, dX):
        pts, inv_T, vol, planes_cl = ctx.saved_tensors
        dvol, dpl = _field_backward(pts, inv_T, vol, planes_cl, ctx.boxes, dX, ctx.needs_input_grad[2], ctx.needs_input_grad[3],
                                    ctx.ray_rows)
        return None, None, dvol, dpl, None, None


def _field_backward(pts, inv_T, vol, planes_cl, boxes, dX, need_vol, need_planes, ray_rows=0):
    """(dvol, dplanes_cl) of the field inputs from dX [n, 2C+48] float32 (hav_field_inputs_bwd, or its fixed-point form).  ray_rows > 0 =
    the samples per ray of queries whose neighbouring rays are neighbouring pixels (hav_field_inputs_bwd_rows: the same sums, the scatter
    merges across 16 rays instead of along one)"""
    p = _field_params(pts, planes_cl, vol, boxes)
    det = deterministic() and p.C <= 64
    mk = torch.empty_like if det else torch.zeros_like          # (the fixed-point route writes every element itself)
    dvol = mk(vol) if need_vol else None
    dpl = mk(planes_cl) if need_planes else None
    if dvol is not None or dpl is not None:
        dX = dX.contiguous()
        L = _lib.lib()
        with torch.cuda.device(pts.device):
            if det:
                # HAVATAR_DETERMINISTIC=1: 64-bit fixed-point sums, integer atomics -- the same bits on every run (the float atomics of
                # the default route land in a different order every time)
                from .conv import absmax
                words = absmax(dX)
                scratch = torch.empty(int(L.hav_field_inputs_bwd_fixed_scratch_bytes(C.byref(p))), dtype=torch.uint8, device=pts.device)
                rc = L.hav_field_inputs_bwd_fixed(_p(dpl), _p(dvol), _p(dX), _p(words), _p(scratch), C.byref(p), _p(pts), _p(inv_T), _p(vol),
                                                  _p(planes_cl), _stream())
            elif ray_rows > 0:
                rc = L.hav_field_inputs_bwd_rows(_p(dpl), _p(dvol), _p(dX), C.byref(p), _p(pts), _p(inv_T), _p(vol), _p(planes_cl), int(ray_rows),
                                                 _stream())
            else:
                rc = L.hav_field_inputs_bwd(_p(dpl), _p(dvol), _p(dX), C.byref(p), _p(pts), _p(inv_T), _p(vol), _p(planes_cl), _stream())
        _lib.check(rc, "hav_field_inputs_bwd")
        from .conv import _trace
        _trace("FieldInputs.bwd dX,dvol,dplanes,vol", dX, dvol, dpl, vol)          # (development aid; a no-op unless HAVATAR_NAN_TRACE)
    return dvol, dpl


class FieldMlp(Function):
    """rf [B*N, 68] = radiance_mlp(field_inputs(pts)) as ONE autograd node: the rows X between the two kernels are bf16 (the MLP's bf16
    kernels round fp32 rows exactly so -- same rf, same gradients as FieldInputs -> FusedMlp), written once and read twice at half the
    bytes, and kept for the backward at half the memory.  Gradients: vol, planes_cl, the ten MLP tensors."""

    @staticmethod
    @_fwd32
    def forward(ctx, pts, inv_T, vol, planes_cl, boxes, ray_rows, *weights):
        from . import mlp_train
        _need_hip("FieldMlp", pts, inv_T, vol, planes_cl)
        pts, inv_T, vol, planes_cl = pts.contiguous(), inv_T.contiguous(), vol.contiguous(), planes_cl.contiguous()
        p = _field_params(pts, planes_cl, vol, boxes)
        X = torch.empty(p.n, 2 * p.C + 48, device=pts.device, dtype=torch.bfloat16)
        with torch.cuda.device(pts.device):
            rc = _lib.lib().hav_field_inputs_fwd_bf16(_p(X), C.byref(p), _p(pts), _p(inv_T), _p(vol), _p(planes_cl), _stream())
        _lib.check(rc, "hav_field_inputs_fwd_bf16")
        blob = mlp_train.pack(weights)
        ctx.save_for_backward(pts, inv_T, vol, planes_cl, X, blob)
        ctx.boxes, ctx.ray_rows = boxes, ray_rows
        ctx.shapes = [tuple(w.shape) for w in weights]
        return mlp_train.forward_only(X, blob)

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, d_rf):
        from . import mlp_train
        pts, inv_T, vol, planes_cl, X, blob = ctx.saved_tensors
        need_vol, need_pl = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dX, grads = mlp_train.backward_only(X, d_rf.contiguous(), blob, ctx.shapes, need_dx=need_vol or need_pl)
        dvol, dpl = _field_backward(pts, inv_T, vol, planes_cl, ctx.boxes, dX, need_vol, need_pl, ctx.ray_rows) if dX is not None else (None, None)
        return (None, None, dvol, dpl, None, None) + tuple(grads)


def field_mlp_eligible(planes_nchw, weights):
    """shapes FieldMlp takes: C = 64 channels per plane (2C + 48 = 176 input columns) and the radiance MLP the bf16 kernels are written for"""
    from . import mlp_train
    return planes_nchw.shape[2] == 64 and [tuple(w.shape) for w in weights] == mlp_train._SHAPES


def field_mlp(pts, inv_T, vol, planes_nchw, nerf_box, skin_box, weights, ray_rows=0):
    """as field_inputs() followed by mlp_train.fused_mlp(): pts [B,N,3] ... -> rf [B*N, 68]"""
    boxes = (tuple(nerf_box[0]), tuple(nerf_box[1]), tuple(skin_box[0]), tuple(skin_box[1]))
    return FieldMlp.apply(pts, inv_T, vol, planes_nchw.permute(0, 1, 3, 4, 2).contiguous(), boxes, int(ray_rows), *weights)


def field_inputs(pts, inv_T, vol, planes_nchw, nerf_box, skin_box, ray_rows=0):
    """pts [B,N,3], inv_T [B,4,3], vol [1,2,D,D,D], planes [2,B,C,H,W] (the Trainer's layout), boxes = (scale3, trans3) of the two
    UniformBoxWarp_new modules -> X [B*N, 2C+48].  The NCHW -> channels-last permutation is a differentiable ATen op.  ray_rows = S when
    pts is [B, rays, S] flattened and neighbouring rays are neighbouring pixels of an image row (the training patch): a hint for the backward."""
    boxes = (tuple(nerf_box[0]), tuple(nerf_box[1]), tuple(skin_box[0]), tuple(skin_box[1]))
    return FieldInputs.apply(pts, inv_T, vol, planes_nchw.permute(0, 1, 3, 4, 2).contiguous(), boxes, int(ray_rows))


class Composite(Function):
    """(rgb [n,CH], acc [n], weights [n,S], depth [n]) = volume_render_radiance_field(rf [n,S,CH+1], z, rd, noise, bg)."""

    @staticmethod
    @_fwd32
    def forward(ctx, rf, z, rd, noise, bg, n_sigmoid):
        _need_hip("Composite", rf, z, rd, noise, bg)
        rf, z, rd = rf.contiguous(), z.contiguous(), rd.contiguous()
        noise = noise.contiguous() if noise is not None else None
        bg = bg.contiguous() if bg is not None else None
        n, S, RW = rf.shape
        if z.shape != (n, S) or rd.shape != (n, 3) or (noise is not None and noise.shape != (n, S)) or (bg is not None and bg.shape != (n, 3)):
            raise RuntimeError("Composite: rf [n,S,CH+1], z [n,S], rd [n,3], noise [n,S], bg [n,3]")
        rgb = torch.empty(n, RW - 1, device=rf.device, dtype=torch.float32)
        acc, depth = torch.empty(n, device=rf.device), torch.empty(n, device=rf.device)
        w = torch.empty(n, S, device=rf.device)
        with torch.cuda.device(rf.device):
            rc = _lib.lib().hav_composite_fwd(_p(rgb), _p(acc), _p(w), _p(depth), _p(rf), _p(z), _p(rd), _p(noise), _p(bg), n, S, RW - 1,
                                              int(n_sigmoid), _stream())
        _lib.check(rc, "hav_composite_fwd")
        ctx.save_for_backward(rf, z, rd, noise, bg)
        ctx.n_sigmoid = int(n_sigmoid)
        return rgb, acc, w, depth

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, d_rgb, d_acc, d_w, d_depth):
        rf, z, rd, noise, bg = ctx.saved_tensors
        n, S, RW = rf.shape
        d_rf = torch.empty_like(rf)
        d_rgb = d_rgb.contiguous() if d_rgb is not None else torch.zeros(n, RW - 1, device=rf.device)
        d_acc, d_w, d_depth = [t.contiguous() if t is not None else None for t in (d_acc, d_w, d_depth)]
        with torch.cuda.device(rf.device):
            rc = _lib.lib().hav_composite_bwd(_p(d_rf), _p(d_rgb), _p(d_acc), _p(d_w), _p(d_depth), _p(rf), _p(z), _p(rd), _p(noise),
                                              _p(bg), n, S, RW - 1, ctx.n_sigmoid, _stream())
        _lib.check(rc, "hav_composite_bwd")
        from .conv import _trace
        _trace("Composite.bwd d_rf,d_rgb,rf", d_rf, d_rgb, rf)
        return d_rf, None, None, None, None, None


def composite(rf, z, rd, noise=None, bg=None, n_sigmoid=3):
    return Composite.apply(rf, z, rd, noise, bg, n_sigmoid)


def resample_depths(z, weights, num_fine, zeta=None, return_samples=False):
    """z2 [n, ceil(S_c/2)+num_fine] = sort(cat(z[:, ::2], sample_pdf(z_mid, weights[:, 1:-1], num_fine))) -- the statements of
    model/nerf_trainer.py:166-170 + utils/nerf_util.py:76-117 of the reference as one launch (hav_resample_depths).  z, weights
    [n, S_c] HIP float32; zeta [n, num_fine] = the raw torch.rand draw (the caller makes it, in the reference's order), None =
    det=True.  No gradient, like the reference's .detach()."""
    _need_hip("resample_depths", z, weights, zeta)
    z, weights = z.detach().contiguous(), weights.detach().contiguous()
    zeta = zeta.detach().contiguous() if zeta is not None else None
    n, S_c = z.shape
    num_fine = int(num_fine)
    if weights.shape != (n, S_c) or (zeta is not None and zeta.shape != (n, num_fine)):
        raise RuntimeError("resample_depths: z, weights [n,S_c], zeta [n,num_fine]")
    z2 = torch.empty(n, (S_c + 1) // 2 + num_fine, device=z.device, dtype=torch.float32)
    zs = torch.empty(n, num_fine, device=z.device, dtype=torch.float32) if return_samples else None
    with torch.cuda.device(z.device):
        rc = _lib.lib().hav_resample_depths(_p(z2), _p(zs), _p(z), _p(weights), _p(zeta), n, S_c, num_fine, _stream())
    _lib.check(rc, "hav_resample_depths")
    return (z2, zs) if return_samples else z2


class EqualLinearFn(Function):
    """y = F.linear(x, W * scale, bias * lr_mul) for x [B <= 8, in]: one launch forward, one backward (hav_equal_linear_*), instead of the
    3 + 6-7 ATen launches of the statement (model/styleUnet.py:128-162 of the reference) -- the modulation layer of every ModulatedConv2d."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, W, bias, scale, lr_mul):
        _need_hip("EqualLinearFn", x, W, bias)
        x, W = x.contiguous(), W.contiguous()
        bias = bias.contiguous() if bias is not None else None
        B, n_in = x.shape
        n_out = W.shape[0]
        if W.shape != (n_out, n_in) or (bias is not None and bias.shape != (n_out,)):
            raise RuntimeError("EqualLinearFn: x [B,in], W [out,in], bias [out]")
        y = torch.empty(B, n_out, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            rc = _lib.lib().hav_equal_linear_fwd(_p(y), _p(x), _p(W), _p(bias), float(scale), float(lr_mul), B, n_in, n_out, _stream())
        _lib.check(rc, "hav_equal_linear_fwd")
        ctx.save_for_backward(x, W)
        ctx.consts = (float(scale), float(lr_mul), bias is not None)
        return y

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        scale, lr_mul, has_bias = ctx.consts
        dy = dy.contiguous()
        B, n_in = x.shape
        n_out = W.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] else None
        db = torch.empty(n_out, device=x.device, dtype=torch.float32) if (has_bias and ctx.needs_input_grad[2]) else None
        with torch.cuda.device(x.device):
            rc = _lib.lib().hav_equal_linear_bwd(_p(dx), _p(dW), _p(db), _p(dy), _p(x), _p(W), scale, lr_mul, B, n_in, n_out, _stream())
        _lib.check(rc, "hav_equal_linear_bwd")
        return dx, dW, db, None, None


def equal_linear_eligible(x, W, bias):
    """training on HIP float32 tensors, a 2-D input of at most 8 rows (the style vectors of a batch), a shape the kernels take"""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[0] <= 8 and W.dtype == torch.float32 and W.dim() == 2
            and x.shape[1] == W.shape[1] and W.shape[1] <= 4096 and W.shape[0] * W.shape[1] <= (1 << 24)
            and (bias is None or bias.dtype == torch.float32) and torch.is_grad_enabled()
            and (x.requires_grad or W.requires_grad or (bias is not None and bias.requires_grad)))


def equal_linear(x, W, bias, scale, lr_mul):
    return EqualLinearFn.apply(x, W, bias, scale, lr_mul)


class Upsample3d2x(Function):
    """nn.Upsample(scale_factor=2, mode='trilinear', align_corners=False) on a float32 HIP tensor [N,C,D,H,W]: one launch forward,
    one backward (hav_upsample3d_2x_*), instead of the ~30 / ~60 ATen launches of the slice-and-lerp statement."""

    @staticmethod
    @_fwd32
    def forward(ctx, x):
        _need_hip("Upsample3d2x", x)
        x = x.contiguous()
        N, Cc, D, H, W = x.shape
        out = torch.empty(N, Cc, 2 * D, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hav_upsample3d_2x_fwd(_p(out), _p(x), N * Cc, D, H, W, _stream()), "hav_upsample3d_2x_fwd")
        ctx.shape = (N, Cc, D, H, W)
        return out

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, g):
        N, Cc, D, H, W = ctx.shape
        g = g.contiguous()
        dx = torch.empty(N, Cc, D, H, W, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().hav_upsample3d_2x_bwd(_p(dx), _p(g), N * Cc, D, H, W, _stream()), "hav_upsample3d_2x_bwd")
        return dx


def upsample3d_2x(x):
    return Upsample3d2x.apply(x)


class Conv3dSmall(Function):
    """nn.Conv3d(kernel 3, padding 1, stride 1) on a small cubic volume [1,C,R,R,R] (R <= 8: the first three layers of VolumeDecoder) as
    matrix products over an explicit patch matrix (hav_im2col3d / hav_col2im3d + three GEMMs): these layers are weight-bound (56 / 14 /
    3.5 MB of filters for 8 / 64 / 512 voxels) and MIOpen / CK take 350 / 200 / 115 us for their forward alone."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, w, b):
        _need_hip("Conv3dSmall", x, w)
        x, w = x.contiguous(), w.contiguous()
        _, Cc, R = x.shape[:3]
        Cout = w.shape[0]
        col = torch.empty(27 * Cc, R ** 3, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().hav_im2col3d(_p(col), _p(x), Cc, R, _stream()), "hav_im2col3d")
        w2 = w.view(Cout, 27 * Cc)
        y = torch.mm(w2, col) if b is None else torch.addmm(b.view(-1, 1), w2, col)
        ctx.save_for_backward(col, w)
        ctx.shape, ctx.has_bias = (Cc, R), b is not None
        return y.view(1, Cout, R, R, R)

    @staticmethod
    @once_differentiable
    @_bwd32
    def backward(ctx, g):
        col, w = ctx.saved_tensors
        Cc, R = ctx.shape
        Cout = w.shape[0]
        g2 = g.contiguous().view(Cout, R ** 3)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dw = torch.mm(g2, col.t()).view_as(w)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.sum(1)
        if ctx.needs_input_grad[0]:
            dcol = torch.mm(w.view(Cout, 27 * Cc).t(), g2)
            dx = torch.empty(1, Cc, R, R, R, device=g.device, dtype=torch.float32)
            with torch.cuda.device(g.device):
                _lib.check(_lib.lib().hav_col2im3d(_p(dx), _p(dcol), Cc, R, _stream()), "hav_col2im3d")
        return dx, dw, db


def conv3d_small_eligible(x, conv):
    w, b = conv.weight, conv.bias
    if not (w.is_cuda and w.device == x.device and w.dtype == torch.float32 and (b is None or (b.dtype == torch.float32 and b.device == x.device))
            and getattr(conv, "padding_mode", "zeros") == "zeros"):
        return False          # anything else stays on nn.Conv3d (a non-fp32 weight made Conv3dSmall raise instead of falling back)
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[0] == 1 and x.shape[2] == x.shape[3] == x.shape[4] and x.shape[2] <= 8
            and tuple(conv.kernel_size) == (3, 3, 3) and tuple(conv.padding) == (1, 1, 1) and tuple(conv.stride) == (1, 1, 1)
            and tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 and os.environ.get("HAVATAR_CONV3D_SMALL", "1") != "0")


def conv3d_small(x, conv):
    return Conv3dSmall.apply(x, conv.weight, conv.bias)


class Demod(Function):
    """d [B,Cout] = rsqrt(sum_i s[b,i]^2 * scale^2 sum_k W[o,i,k]^2 + eps): the demodulation factors of a ModulatedConv2d
    (reference model/styleUnet.py:214-227, factored form) as one autograd node -- two launches each way (hav_demod_fwd / _bwd)
    instead of ~8 + ~14 ATen launches, three of which stream the whole weight tensor."""

    @staticmethod
    @_fwd32
    def forward(ctx, s, W, scale, eps):
        _need_hip("Demod", s, W)
        s, W = s.contiguous(), W.contiguous()
        B, Cin = s.shape
        Cout, KK = W.shape[0], W.shape[2] * W.shape[3]
        d = torch.empty(B, Cout, device=s.device, dtype=torch.float32)
        q = torch.empty(Cin, Cout, device=s.device, dtype=torch.float32)
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().hav_demod_fwd(_p(d), _p(q), _p(s), _p(W), float(scale), float(eps), B, Cin, Cout, KK, _stream()), "hav_demod_fwd")
        ctx.save_for_backward(s, W, d, q)
        ctx.scale, ctx.eps = float(scale), float(eps)
        return d

    @staticmethod
    @_bwd32
    def backward(ctx, gd):
        s, W, d, q = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True (path-length regulariser through the generator, reference utils/styleUnet_util.py:92): restate the node
            # with ATen under autograd so that the second-order graph exists
            with torch.enable_grad():
                wsq = (ctx.scale * W).pow(2).sum((2, 3)).t()
                dd = torch.rsqrt(torch.matmul(s * s, wsq) + ctx.eps)
                need = [t for t, n in zip((s, W), ctx.needs_input_grad[:2]) if n and t.requires_grad]
                got = iter(torch.autograd.grad(dd, need, gd, create_graph=True) if need else ())
            return tuple(next(got) if (n and t.requires_grad) else None for t, n in zip((s, W), ctx.needs_input_grad[:2])) + (None, None)
        B, Cin = s.shape
        Cout, KK = W.shape[0], W.shape[2] * W.shape[3]
        gs, gW, gq = torch.empty_like(s), torch.empty_like(W), torch.empty_like(q)
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().hav_demod_bwd(_p(gs), _p(gW), _p(gq), _p(gd.contiguous()), _p(s), _p(d), _p(q), _p(W), ctx.scale, B, Cin, Cout,
                                                KK, _stream()), "hav_demod_bwd")
        return gs, gW, None, None


def demod(s, W, scale, eps=1e-8):
    """s [B,Cin] (differentiable), W [Cout,Cin,k,k] raw parameter -> d [B,Cout]."""
    return Demod.apply(s, W, scale, eps)
