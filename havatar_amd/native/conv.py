"""The convolutions of the StyleGAN blocks (reference model/styleUnet.py:165-297,326-368,565-599) on the fp16 matrix cores with split
operands (libhavatar_hip.so: hav_conv3x3_*, hav_gemm_*, hav_upconv_finish; include/havatar.h):
  conv3x3 / pack            3x3 stride 1 with the block's glue fused in (inference)
  upconv3x3 / pack_upconv   the up-sampling StyledConv: transposed convolution as a matrix product + scatter / blur / epilogue
  conv3x3_autograd          forward, data gradient (same kernel) and weight gradient (wgrad3x3 = hav_conv3x3_wgrad) for training
HIP float32 tensors only; no fallback in here -- the `*eligible()` predicates tell the caller whether a shape is supported, otherwise
it keeps its MIOpen route."""
import ctypes as C
import os

import torch

from .. import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _c(t):
    return None if t is None else t.contiguous()


def _stream(device):
    """the current stream OF `device` (not of the current device)"""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _autoscale_default():
    return os.environ.get("HAVATAR_CONV_AUTOSCALE", "1") != "0"


# Development aid (tools/graph_anomaly_hunt.py, tests/test_harness.py): is-finite flags per operand / result of the wrappers below, evaluated
# INSIDE a captured step at every replay.  HAVATAR_NAN_TRACE=1: torch.isfinite(t).all() -- two ATen launches and two temporaries per tensor in
# the graph's memory pool.  HAVATAR_NAN_TRACE=2 (what the tests use): one hav_debug_nonfinite launch per tensor into a flag buffer that is
# allocated OUTSIDE any graph pool -- no temporaries; the buffer is zeroed by nan_trace_reset() before a step / replay.
_NAN_MODE = int(os.environ.get("HAVATAR_NAN_TRACE", "0") or 0)
_NAN_TRACE = [] if _NAN_MODE else None
_NAN_BUF = {}          # device -> int32 flag buffer (mode 2)
_NAN_BUF_WORDS = 8192


class _Flag:
    """truthy iff the traced tensor was finite (mode 2: word i of the flag buffer is still 0)"""

    def __init__(self, buf, i):
        self.buf, self.i = buf, i

    def __bool__(self):
        return int(self.buf[self.i]) == 0


def _nan_buf(device):
    b = _NAN_BUF.get(device)
    if b is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("nan trace: allocate the flag buffer before the capture (nan_trace_reset(device))")
        b = _NAN_BUF[device] = torch.zeros(_NAN_BUF_WORDS, dtype=torch.int32, device=device)
    return b


def nan_trace_reset(device):
    """zero the flag words on the current stream (before an eager step or a graph replay); allocates the buffer on first use"""
    if _NAN_TRACE is not None and _NAN_MODE != 1:
        _nan_buf(torch.device(device)).zero_()


def _trace(name, *ts):
    if _NAN_TRACE is None:
        return
    for k, t in enumerate(ts):
        if t is None:
            continue
        label = "%s[%d]%s" % (name, k, tuple(t.shape))
        if _NAN_MODE == 1:
            _NAN_TRACE.append((label, torch.isfinite(t if t.is_floating_point() else t.view(torch.float32)).all()))
            continue
        if not t.is_contiguous() or t.element_size() != 4:
            continue
        buf, i = _nan_buf(t.device), len(_NAN_TRACE)
        if i >= _NAN_BUF_WORDS:
            continue
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib().hav_debug_nonfinite(C.c_void_p(buf.data_ptr() + 4 * i), _p(t), t.numel(), _stream(t.device)), "hav_debug_nonfinite")
        _NAN_TRACE.append((label, _Flag(buf, i)))


def _absmax(t, st):
    """HAV_ABSMAX_WORDS partial maxima of |t| (hav_absmax: one pass, no host round trip) for the kernels' power-of-two range control"""
    words = torch.empty(256, dtype=torch.int32, device=t.device)
    _lib.check(_lib.lib().hav_absmax(_p(words), _p(t), t.numel(), st), "hav_absmax")
    return words


def absmax(t):
    """the HAV_ABSMAX_WORDS partial maxima of |t| on t's current stream (what `amax` / `g_amax` / `x_amax` arguments take)"""
    with torch.cuda.device(t.device):
        return _absmax(t.contiguous(), _stream(t.device))


def eligible(x, weight, stride=1, padding=1):
    """weight [Cout,Cin,3,3]; x [B,Cin,H,W] float32 on a HIP device."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    Cout, Cin, kh, kw = weight.shape
    B, Ci, H, W = x.shape
    # tiles: 4 rows x 32 columns, or 8 x 16 for maps 16 (48, 80, ...) columns wide (the 16^2 layers)
    return (kh == 3 and kw == 3 and stride == 1 and padding == 1 and Ci == Cin and Cin % 16 == 0 and Cout % 64 == 0
            and ((H % 4 == 0 and W % 32 == 0) or (H % 8 == 0 and W % 16 == 0)))


def pack(weight, wmul=1.0):
    """[Cout,Cin,3,3] float32 -> fragment blob (uint8 tensor) with `wmul` folded in."""
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[:2]
    L = _lib.lib()
    blob = torch.empty(int(L.hav_conv3x3_packed_bytes(Cout, Cin)), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.hav_conv3x3_pack(_p(blob), _p(w), Cout, Cin, float(wmul), _stream(w.device)), "hav_conv3x3_pack")
    return blob


def pack_t(weight, wmul=1.0):
    """[Cout,Cin,3,3] float32 -> the fragment blob of the DATA GRADIENT's convolution (Cin outputs, Cout inputs, flipped taps), packed
    straight from `weight` (hav_conv3x3_pack_t): no flip / transpose / contiguous passes."""
    w = weight.detach().contiguous()
    Cout, Cin = w.shape[:2]
    L = _lib.lib()
    blob = torch.empty(int(L.hav_conv3x3_packed_bytes(Cin, Cout)), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.hav_conv3x3_pack_t(_p(blob), _p(w), Cout, Cin, float(wmul), _stream(w.device)), "hav_conv3x3_pack_t")
    return blob


def conv1x1_eligible(x, weight):
    """weight [Cout,Cin,1,1]; x [B,Cin,H,W] float32 on a HIP device: shapes hav_gemm_split takes as y[Cout, HW] = W . x[Cin, HW]."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    Cout, Cin, kh, kw = weight.shape
    return kh == 1 and kw == 1 and x.shape[1] == Cin and Cin % 32 == 0 and (x.shape[2] * x.shape[3]) % 128 == 0


def pack_1x1(weight, wmul=1.0):
    """[Cout,Cin,1,1] -> fragment blob of the [Cout x Cin] matrix (hav_gemm_pack) with `wmul` folded in."""
    w = weight.detach()
    Cout, Cin = w.shape[:2]
    a = w.reshape(Cout, Cin).contiguous()
    L = _lib.lib()
    blob = torch.empty(int(L.hav_gemm_packed_bytes(Cout, Cin)), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.hav_gemm_pack(_p(blob), _p(a), Cout, Cin, float(wmul), _stream(w.device)), "hav_gemm_pack")
    return blob


def conv1x1(x, packed, Cout, autoscale=None):
    """y [B,Cout,H,W] = conv2d(x, W [Cout,Cin,1,1]): the 1x1 EqualConv2d of ConvLayer / FromRGB (reference model/styleUnet.py:104-122,
    251-266) as one split-fp16 matrix product per sample (hav_gemm_split) instead of rocBLAS' fp32 GEMM; bias / activation stay with the
    caller (FusedLeakyReLU = hav_fused_bias_act)."""
    if autoscale is None:
        autoscale = _autoscale_default()
    x = x.contiguous()
    B, Cin, H, W = x.shape
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        amax = _absmax(x, st) if autoscale else None
        _lib.check(L.hav_gemm_split(_p(y), _p(x), _p(packed), None, _p(amax), B, Cout, Cin, H * W, st), "hav_gemm_split")
    return y


def conv3x3(x, packed, Cout, s=None, d=None, noise=None, noise_weight=None, bias=None, slope=0.2, gain=2 ** 0.5, act=True, autoscale=None, amax=None):
    """y = act(d * conv3x3(s * x, W) + noise_weight * noise + bias) * gain; see include/havatar.h for the exact order.
    autoscale (default on; HAVATAR_CONV_AUTOSCALE=0 turns the default off): s * x is brought into fp16's comfortable range by an
    exact power-of-two scale found on the device (hav_absmax: one pass over x; the kernel multiplies the maximum by max |s_b|) and
    undone in the epilogue -- gradients (1e-6) keep their low parts, activations and modulations of any size cannot overflow the
    fp16 split.  ~1 % of a frame.  amax: the words of absmax(x) when the caller already has them (the training nodes use an operand's
    maxima for its forward, data-gradient and weight-gradient launches)."""
    if autoscale is None:
        autoscale = _autoscale_default() or amax is not None
    x = x.contiguous()
    B, Cin, H, W = x.shape
    y = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    nb = 0
    if noise is not None:
        noise = noise.contiguous()
        if noise.numel() == B * H * W and B > 1:
            nb = 1
        elif noise.numel() != H * W:
            raise RuntimeError("conv3x3: noise must be [1,1,H,W] or [B,1,H,W]")
    s, d, bias, noise_weight = _c(s), _c(d), _c(bias), _c(noise_weight)
    L = _lib.lib()
    need = int(L.hav_conv3x3_scratch_bytes(B, Cin, Cout, H, W))          # small maps: K-split slices, summed in a second pass
    scratch = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        if amax is None:
            amax = _absmax(x, st) if autoscale else None
        rc = L.hav_conv3x3_split(_p(y), _p(x), _p(packed), _p(s), _p(d), _p(noise), _p(noise_weight), _p(bias), float(slope),
                                 float(gain), int(bool(act)), nb, B, Cin, Cout, H, W, _p(scratch), _p(amax), st)
    _lib.check(rc, "hav_conv3x3_split")
    _trace("conv3x3 x,amax,y", x, amax, y)
    return y


def s2_eligible(x, weight, stride=2, padding=0, shape=None):
    """weight [Cout,Cin,3,3]; x [B,Cin,Hin,Win] float32 on a HIP device: shapes hav_conv3x3s2_split takes.  shape: the convolution's input
    shape when x is the tensor in front of a Blur that has not run yet."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    Cout, Cin, kh, kw = weight.shape
    B, Ci, H, W = x.shape if shape is None else shape
    if not (kh == 3 and kw == 3 and stride == 2 and padding in (0, 1) and Ci == Cin and Cin % 16 == 0 and Cout % 64 == 0):
        return False
    Ho, Wo = (H + 2 * padding - 3) // 2 + 1, (W + 2 * padding - 3) // 2 + 1
    return H >= 3 and W >= 3 and Ho % 4 == 0 and Wo % 32 == 0 and Cin * H * W < 2 ** 31


def conv3x3s2(x, packed, Cout, padding=0, s=None, d=None, noise=None, noise_weight=None, bias=None, slope=0.2, gain=2 ** 0.5, act=True,
              autoscale=None):
    """y = act(d * conv3x3(s * x, W, stride 2, padding) + noise_weight * noise + bias) * gain (hav_conv3x3s2_split; `packed` from pack()):
    the down-sampling EqualConv2d of ConvLayer / ConvBlock (model/styleUnet.py:326-368) with bias + leaky-ReLU fused."""
    if autoscale is None:
        autoscale = _autoscale_default()
    x = x.contiguous()
    B, Cin, H, W = x.shape
    Ho, Wo = (H + 2 * padding - 3) // 2 + 1, (W + 2 * padding - 3) // 2 + 1
    y = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=x.device)
    nb = 0
    if noise is not None:
        noise = noise.contiguous()
        if noise.numel() == B * Ho * Wo and B > 1:
            nb = 1
        elif noise.numel() != Ho * Wo:
            raise RuntimeError("conv3x3s2: noise must be [1,1,Hout,Wout] or [B,1,Hout,Wout]")
    s, d, bias, noise_weight = _c(s), _c(d), _c(bias), _c(noise_weight)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        amax = _absmax(x, st) if autoscale else None
        nbytes = int(L.hav_conv3x3s2_scratch_bytes(B, Cin, Cout, H, W, int(padding)))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        rc = L.hav_conv3x3s2_split(_p(y), _p(x), _p(packed), _p(s), _p(d), _p(noise), _p(noise_weight), _p(bias), float(slope), float(gain),
                                   int(bool(act)), nb, B, Cin, Cout, H, W, int(padding), _p(scratch), _p(amax), st)
    _lib.check(rc, "hav_conv3x3s2_split")
    _trace("conv3x3s2 x,amax,y", x, amax, y)
    return y


def wgrad_eligible(g, x):
    if not (g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.dim() == 4 and x.dim() == 4):
        return False
    B, Cout, H, W = g.shape
    return x.shape[0] == B and tuple(x.shape[2:]) == (H, W) and x.shape[1] % 32 == 0 and Cout % 64 == 0 and W % 16 == 0


def wgrad3x3(g, x, xs=None, out_mul=1.0, g_amax=None, x_amax=None):
    """gw [Cout,Cin,3,3] = out_mul * d/dW of conv2d(xs * x, W, stride 1, padding 1) given g = dL/dy (hav_conv3x3_wgrad_mod: split-fp16 MFMA,
    fp32-class; both operands under the power-of-two range control: g is gradient-sized, x whatever the activations are).  xs [B,Cin]:
    the modulation of a ModulatedConv2d (None: plain)."""
    g, x, xs = g.contiguous(), x.contiguous(), _c(xs)
    B, Cout, H, W = g.shape
    Cin = x.shape[1]
    L = _lib.lib()
    gw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=g.device)
    scratch = torch.empty(int(L.hav_conv3x3_wgrad_scratch_bytes(B, Cin, Cout, H, W)), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        st = _stream(g.device)
        g_amax = _absmax(g, st) if g_amax is None else g_amax
        x_amax = _absmax(x, st) if x_amax is None else x_amax
        _lib.check(L.hav_conv3x3_wgrad_mod(_p(gw), _p(g), _p(x), _p(xs), float(out_mul), _p(scratch), _p(g_amax), _p(x_amax), B, Cin, Cout,
                                           H, W, st), "hav_conv3x3_wgrad_mod")
    return gw


def wgrad_s2_eligible(S, L):
    """S [B,M,H,W], L [B,N,2H+1,2W+1]: shapes hav_conv3x3s2_wgrad takes (HAVATAR_S2_WGRAD=0 keeps ATen's convolution_backward)."""
    if not (S.is_cuda and S.dtype == torch.float32 and L.dtype == torch.float32 and S.dim() == 4 and L.dim() == 4):
        return False
    B, M, H, W = S.shape
    N = L.shape[1]
    if L.shape[0] != B or tuple(L.shape[2:]) != (2 * H + 1, 2 * W + 1) or os.environ.get("HAVATAR_S2_WGRAD", "1") == "0":
        return False
    if W < 16:          # the small-map kernel (4^2 / 8^2 layers): plain fp32, its operand rows must fit in LDS
        return M % 16 == 0 and N % 16 == 0 and (16 * B * H * W + 16 * ((B * (2 * H + 1) * (2 * W + 1)) | 1)) * 4 <= 64 * 1024
    return M % 64 == 0 and N % 32 == 0 and W % 16 == 0


def wgrad3x3s2(S, L, ss=None, out_mul=1.0, transpose=False, s_amax=None, l_amax=None):
    """out[m,n,ky,kx] = out_mul * sum_{b,y,x} ss[b,m] * S[b,m,y,x] * L[b,n,2y+ky,2x+kx] (hav_conv3x3s2_wgrad, include/havatar.h): the weight
    gradient of both stride-2 layers of the StyleGAN blocks -- the down-sampling ConvLayer (S = dL/dy, L = the blurred input -> [Cout,Cin,3,3])
    and the up-sampling StyledConv's transposed convolution (S = x, ss = its modulation, L = dL/d(conv_transpose2d output), transpose=True ->
    the parameter's [Cout,Cin,3,3]).  Both operands under the power-of-two range control."""
    S, L = S.contiguous(), L.contiguous()
    B, M, H, W = S.shape
    N = L.shape[1]
    lib = _lib.lib()
    out = torch.empty((N, M, 3, 3) if transpose else (M, N, 3, 3), dtype=torch.float32, device=S.device)
    nscr = int(lib.hav_conv3x3s2_wgrad_scratch_bytes(B, M, N, H, W))
    scratch = torch.empty(nscr, dtype=torch.uint8, device=S.device) if nscr else None
    ss = _c(ss)
    with torch.cuda.device(S.device):
        st = _stream(S.device)
        if W >= 16:          # (the small-map kernel multiplies in fp32: no range control)
            s_amax = _absmax(S, st) if s_amax is None else s_amax
            l_amax = _absmax(L, st) if l_amax is None else l_amax
        _lib.check(lib.hav_conv3x3s2_wgrad(_p(out), _p(S), _p(L), _p(ss), float(out_mul), int(bool(transpose)), _p(scratch), _p(s_amax), _p(l_amax),
                                           B, M, N, H, W, st), "hav_conv3x3s2_wgrad")
    return out


class _Conv3x3Split(torch.autograd.Function):
    """conv2d(x, w, stride 1, padding 1) for training: the forward runs on hav_conv3x3_split (split-fp16 MFMA, fp32-class results,
    ~2x MIOpen's fp32 Winograd on the encoder shapes), and so does the data gradient, which is the same kind of convolution with
    the transposed, flipped filters; the weight gradient runs on hav_conv3x3_wgrad (wgrad3x3 below; ATen's convolution_backward only
    for shapes that kernel does not take, HAVATAR_CONV_WGRAD=0 forces it).
    Double backward (create_graph=True: the R1 and path-length regularisers of stage two, reference utils/styleUnet_util.py:74,92):
    the backward then runs under grad mode and states both gradients with ATen's differentiable convolution ops instead of the
    kernels -- slower, but the second-order graph is complete.  The weight gradient is skipped inside
    conv2d_gradfix.no_weight_gradients(), as the reference's Conv2d backward does (model/op/conv2d_gradfix.py:155)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return conv3x3(x, pack(w, 1.0), w.shape[0], act=False)

    @staticmethod
    def backward(ctx, g):
        from ..model.op import conv2d_gradfix
        x, w = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled
        if torch.is_grad_enabled():          # create_graph=True: differentiable statements of both gradients
            gx = torch.nn.grad.conv2d_input(x.shape, w, g, stride=1, padding=1) if need_x else None
            gw = torch.nn.grad.conv2d_weight(x, w.shape, g, stride=1, padding=1) if need_w else None
            return gx, gw
        g = g.contiguous()
        gx = None
        if need_x and os.environ.get("HAVATAR_CONV_BWD", "1") != "0":
            # dL/dx is itself a 3x3 / stride 1 / padding 1 convolution of g with the transposed, flipped filters: the same kernel
            wt = w.flip(2, 3).transpose(0, 1).contiguous()
            if eligible(g, wt):
                gx = conv3x3(g, pack(wt, 1.0), wt.shape[0], act=False, autoscale=True)     # gradients are ~1e-6: see hav_absmax
                need_x = False
        gw = None
        if need_w and os.environ.get("HAVATAR_CONV_WGRAD", "1") != "0" and wgrad_eligible(g, x):
            gw = wgrad3x3(g, x)          # the weight gradient on the split-fp16 path too (MIOpen: fp32 igemm_wrw + NHWC transposes)
            need_w = False
        if not (need_x or need_w):
            return gx, gw
        r = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [need_x, need_w, False])
        return (gx if gx is not None else r[0]), (gw if gw is not None else r[1])


def conv3x3_autograd(x, w):
    """x [B,Cin,H,W], w [Cout,Cin,3,3] (already scaled): differentiable 3x3 / stride 1 / padding 1 convolution, see _Conv3x3Split."""
    return _Conv3x3Split.apply(x.contiguous(), w.contiguous())


def block_eligible(x, weight):
    """x [B,Cin,H,W], weight [Cout,Cin,3,3]: shapes for which the forward, the data gradient (the transposed convolution: Cin and Cout
    swap roles) and the weight gradient of fused_block all run on this library's kernels."""
    if not eligible(x, weight):
        return False
    Cout, Cin = weight.shape[:2]
    return Cin % 64 == 0 and Cout % 64 == 0 and os.environ.get("HAVATAR_FUSED_BLOCK", "1") != "0"


class _FusedConvBlock(torch.autograd.Function):
    """y = act(d * conv3x3(s * x, scale * W) + nw * noise + bias) * gain -- a whole StyledConv / ConvLayer (reference
    model/styleUnet.py:165-310,326-368,565-599) as ONE autograd node: forward = pack + range control + hav_conv3x3_split; backward =
    hav_conv_block_bwd (activation gradient, the d / bias / noise-weight gradients) -> the data gradient on hav_conv3x3_split with filters
    packed by hav_conv3x3_pack_t -> hav_mod_input_bwd (s gradient, input scaling) -> hav_conv3x3_wgrad_mod (weight PARAMETER gradient).
    About a dozen launches where the unfused statement under autograd ran ~40 (scale, modulate, demodulate, noise, bias, activation and
    their gradients as separate ATen kernels).  W is the raw parameter; `scale` is folded into the packs and the weight gradient.
    Under create_graph=True the backward restates the block with differentiable ATen ops; inside
    conv2d_gradfix.no_weight_gradients() no weight gradient is formed."""

    @staticmethod
    def forward(ctx, x, W, s, d, noise, nw, bias, scale, slope, gain, act):
        # (x and W arrive contiguous: fused_block() makes them so OUTSIDE the node -- a .contiguous() in here, where grad mode is off, would
        # save a detached copy of a channels-last / sliced input and cut the second-order graph of the create_graph branch below)
        ax = absmax(x)          # one pass over x serves the forward and, in backward, the weight gradient
        # (HAVATAR_CONV_AUTOSCALE=0, the A/B switch of the range control, reaches this forward too: the words are then only kept for the
        # weight gradient, whose range control is not optional -- gradients of 1e-6 lose their low parts without it)
        y = conv3x3(x, pack(W, scale), W.shape[0], s=s, d=d, noise=noise, noise_weight=nw, bias=bias, slope=slope, gain=gain, act=act,
                    amax=ax if _autoscale_default() else None)
        ctx.save_for_backward(x, W, s, d, noise, nw, bias, y, ax)
        ctx.cfg = (float(scale), float(slope), float(gain), bool(act))
        return y

    @staticmethod
    def backward(ctx, g):
        from ..model.op import conv2d_gradfix
        x, W, s, d, noise, nw, bias, y, ax = ctx.saved_tensors
        scale, slope, gain, act = ctx.cfg
        need = ctx.needs_input_grad
        B, Cout, H, Wd = y.shape
        Cin = x.shape[1]
        if torch.is_grad_enabled():          # create_graph=True: the block as differentiable ATen ops
            with torch.enable_grad():
                v = torch.nn.functional.conv2d(x * s.view(B, Cin, 1, 1) if s is not None else x, W * scale, padding=1)
                if d is not None:
                    v = v * d.view(B, Cout, 1, 1)
                if noise is not None:
                    v = v + nw * noise
                if bias is not None:
                    v = v + bias.view(1, -1, 1, 1)
                if act:
                    v = torch.nn.functional.leaky_relu(v, slope) * gain
                slots = {0: x, 1: W, 2: s, 3: d, 5: nw, 6: bias}
                if conv2d_gradfix.weight_gradients_disabled:
                    slots.pop(1)
                idx = [k for k, t in slots.items() if t is not None and need[k]]          # by ctx.needs_input_grad alone
                got = torch.autograd.grad(v, [slots[k] for k in idx], g, create_graph=True, allow_unused=True) if idx else ()
            out = [None] * 11
            for k, gk in zip(idx, got):
                out[k] = gk
            return tuple(out)
        g = g.contiguous()
        L = _lib.lib()
        dev = y.device
        gc = torch.empty_like(y)
        sums = torch.empty(B * Cout * 3, dtype=torch.float32, device=dev)
        gd = torch.empty(B, Cout, dtype=torch.float32, device=dev) if (d is not None and need[3]) else None
        gb = torch.empty(Cout, dtype=torch.float32, device=dev) if (bias is not None and need[6]) else None
        gnw = torch.empty(1, dtype=torch.float32, device=dev) if (noise is not None and nw is not None and need[5]) else None
        nb = 1 if (noise is not None and B > 1 and noise.numel() == B * H * Wd) else 0
        with torch.cuda.device(dev):
            st = _stream(dev)
            _lib.check(L.hav_conv_block_bwd(_p(gc), _p(gd), _p(gb), _p(gnw), _p(sums), _p(g), _p(y), _p(d), _p(noise), _p(nw), _p(bias), slope, gain,
                                            int(act), nb, B, Cout, H * Wd, st), "hav_conv_block_bwd")
        gx = gs = None
        want_w = need[1] and not conv2d_gradfix.weight_gradients_disabled
        ag = absmax(gc) if (need[0] or (s is not None and need[2]) or want_w) else None          # one pass: data AND weight gradient
        if need[0] or (s is not None and need[2]):
            gx = conv3x3(gc, pack_t(W, scale), Cin, act=False, amax=ag)          # dL/d(s x): gradient-sized, see hav_absmax
            if s is not None:
                gs = torch.empty(B, Cin, dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(L.hav_mod_input_bwd(_p(gx), _p(gs), _p(x), _p(s), B, Cin, H * Wd, _stream(dev)), "hav_mod_input_bwd")
        gW = None
        if want_w:
            gW = wgrad3x3(gc, x, xs=s, out_mul=scale, g_amax=ag, x_amax=ax)
        if gnw is not None and nw.shape != gnw.shape:
            gnw = gnw.view(nw.shape)
        if gb is not None and bias.shape != gb.shape:
            gb = gb.view(bias.shape)
        _trace("FusedConvBlock.bwd g,gc,gx,gs,gW,gd,gnw,gb", g, gc, gx, gs, gW, gd, gnw, gb)
        return (gx if need[0] else None), gW, (gs if need[2] else None), gd, None, gnw, gb, None, None, None, None


def fused_block(x, W, scale, s=None, d=None, noise=None, noise_weight=None, bias=None, slope=0.2, gain=2 ** 0.5, act=True):
    """see _FusedConvBlock; x [B,Cin,H,W], W [Cout,Cin,3,3] raw parameter, s [B,Cin], d [B,Cout], noise [1|B,1,H,W] (not differentiated),
    noise_weight [1], bias [Cout]; callers check block_eligible(x, W)."""
    f = lambda t: None if t is None else t.contiguous()
    if noise is not None and noise_weight is None:
        noise = None
    return _FusedConvBlock.apply(x.contiguous(), W.contiguous(), f(s), f(d), f(noise), f(noise_weight), f(bias), scale, slope, gain, act)


class _S2ConvBlock(torch.autograd.Function):
    """y = act(conv3x3(blur(x), scale * W, stride 2, padding) + bias) * gain -- a down-sampling ConvLayer (Blur -> EqualConv2d stride 2 ->
    FusedLeakyReLU, reference model/styleUnet.py:326-368) as ONE autograd node.  Forward = hav_upfirdn2d + pack + range control +
    hav_conv3x3s2_split instead of the weight scaling, MIOpen's Im2d2Col + fp32 GEMM and the activation launch.  Backward:
    hav_conv_block_bwd (activation and bias gradients in one pass); the data gradient THROUGH THE BLUR is conv_transpose2d(g, W, stride 2)
    followed by the blur's adjoint, a 4x4 FIR with padding (1, 1) -- which is the up-sampling StyledConv's forward, so it runs on
    hav_gemm_split + hav_upconv_finish with Cin and Cout in swapped roles (where their shapes allow; else ATen + hav_upfirdn2d); the weight
    gradient is ATen's (MIOpen) on the saved blurred map.  `fir` None: x is taken as already blurred.  Under create_graph=True the backward
    restates the layer with differentiable ATen ops; inside conv2d_gradfix.no_weight_gradients() no weight gradient is formed."""

    @staticmethod
    def forward(ctx, x, W, bias, fir, scale, slope, gain, act, padding, fpad):
        from ..model.op.upfirdn2d import upfirdn2d as _ufd
        xb = _ufd(x, fir, pad=fpad) if fir is not None else x
        if os.environ.get("HAVATAR_S2_TRAIN_FWD", "kernel") == "aten":          # (A/B switch: the forward on ATen / MIOpen)
            y = torch.nn.functional.conv2d(xb, W * scale, stride=2, padding=padding)
            if bias is not None:
                y = y + bias.view(1, -1, 1, 1)
            if act:
                y = torch.nn.functional.leaky_relu(y, slope) * gain
        else:
            y = conv3x3s2(xb, pack(W, scale), W.shape[0], padding, bias=bias, slope=slope, gain=gain, act=act)
        ctx.save_for_backward(x, xb, W, bias, fir, y)          # (x itself is only read under create_graph; its producer keeps it alive anyway)
        ctx.cfg = (float(scale), float(slope), float(gain), bool(act), int(padding), tuple(fpad) if fpad is not None else None, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        from ..model.op import conv2d_gradfix
        from ..model.op.upfirdn2d import upfirdn2d as _ufd
        x, xb, W, bias, fir, y = ctx.saved_tensors
        scale, slope, gain, act, padding, fpad, xshape = ctx.cfg
        need = ctx.needs_input_grad
        B, Cout, Ho, Wo = y.shape
        Cin = W.shape[1]
        if torch.is_grad_enabled():          # create_graph=True: the layer as differentiable ATen ops
            with torch.enable_grad():
                v = x
                if fir is not None:
                    kh, kw = fir.shape
                    p = fpad if len(fpad) == 4 else (fpad[0], fpad[1], fpad[0], fpad[1])
                    v = torch.nn.functional.conv2d(torch.nn.functional.pad(v, p), fir.flip(0, 1).view(1, 1, kh, kw).expand(Cin, 1, kh, kw), groups=Cin)
                v = torch.nn.functional.conv2d(v, W * scale, stride=2, padding=padding)
                if bias is not None:
                    v = v + bias.view(1, -1, 1, 1)
                if act:
                    v = torch.nn.functional.leaky_relu(v, slope) * gain
                slots = {0: x, 1: W, 2: bias}
                if conv2d_gradfix.weight_gradients_disabled:
                    slots.pop(1)
                idx = [k for k, t in slots.items() if t is not None and need[k]]
                got = torch.autograd.grad(v, [slots[k] for k in idx], g, create_graph=True, allow_unused=True) if idx else ()
            out = [None] * 10
            for k, gk in zip(idx, got):
                out[k] = gk
            return tuple(out)
        g = g.contiguous()
        dev = y.device
        gc = torch.empty_like(y)
        sums = torch.empty(B * Cout * 3, dtype=torch.float32, device=dev)
        gb = torch.empty(Cout, dtype=torch.float32, device=dev) if (bias is not None and need[2]) else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().hav_conv_block_bwd(_p(gc), None, _p(gb), None, _p(sums), _p(g), _p(y), None, None, None, _p(bias), slope, gain,
                                                    int(act), 0, B, Cout, Ho * Wo, _stream(dev)), "hav_conv_block_bwd")
        want_w = need[1] and not conv2d_gradfix.weight_gradients_disabled
        gx = gW = None
        if need[0]:
            wt = W.transpose(0, 1).contiguous()          # [Cin,Cout,3,3]: as an up-sampling layer's parameter it maps Cout -> Cin channels
            fused = (fir is not None and tuple(fir.shape) == (4, 4) and tuple(fpad) in ((2, 2), (2, 2, 2, 2)) and padding == 0
                     and tuple(xshape[2:]) == (2 * Ho, 2 * Wo) and upconv_eligible(gc, wt) and os.environ.get("HAVATAR_S2_DGRAD", "1") != "0")
            if fused:
                gx = upconv3x3(gc, pack_upconv(wt, scale), Cin, fir.flip(0, 1), act=False, autoscale=True)          # gradient-sized: see hav_absmax
            else:
                gx, _, _ = torch.ops.aten.convolution_backward(gc, xb, W * scale, None, [2, 2], [padding, padding], [1, 1], False, [0, 0], 1,
                                                               [True, False, False])
                if fir is not None:
                    kh, kw = fir.shape
                    p = fpad if len(fpad) == 4 else (fpad[0], fpad[1], fpad[0], fpad[1])
                    gx = _ufd(gx, fir.flip(0, 1), pad=(kw - 1 - p[0], kw - 1 - p[1], kh - 1 - p[2], kh - 1 - p[3]))
        if want_w:
            if padding == 0 and wgrad_s2_eligible(gc, xb):
                gW = wgrad3x3s2(gc, xb, out_mul=scale)          # split-fp16 MFMA kernel (MIOpen: fp32 igemm_wrw + NHWC transposes)
            else:
                _, gW, _ = torch.ops.aten.convolution_backward(gc, xb, W, None, [2, 2], [padding, padding], [1, 1], False, [0, 0], 1, [False, True, False])
                gW = gW * scale
        if gb is not None and bias.shape != gb.shape:
            gb = gb.view(bias.shape)
        _trace("S2ConvBlock.bwd g,gc,gx,gW,gb", g, gc, gx, gW, gb)
        return gx, gW, gb, None, None, None, None, None, None, None


def s2_block(x, W, scale, bias=None, slope=0.2, gain=2 ** 0.5, act=True, padding=0, fir=None, fir_pad=None):
    """see _S2ConvBlock; x [B,Cin,H,W]; fir / fir_pad: the layer's Blur (kernel [kh,kw], pad (p0,p1) or (x0,x1,y0,y1)), or None when x is
    already blurred; W [Cout,Cin,3,3] raw parameter.  Callers check s2_eligible on the BLURRED shape."""
    return _S2ConvBlock.apply(x.contiguous(), W.contiguous(), None if bias is None else bias.contiguous(),
                              None if fir is None else fir.detach().contiguous(), scale, slope, gain, act, padding, fir_pad)


def upconv_eligible(x, weight):
    """weight [Cout,Cin,3,3] (the ModulatedConv2d parameter); x [B,Cin,H,W] float32 on a HIP device: shapes hav_gemm_split +
    hav_upconv_finish take."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    Cout, Cin, kh, kw = weight.shape
    B, Ci, H, W = x.shape
    return kh == 3 and kw == 3 and Ci == Cin and Cin % 32 == 0 and (H * W) % 128 == 0 and (19 * (2 * W + 4) * 4) <= 64 * 1024


def pack_upconv(weight, wmul=1.0):
    """[Cout,Cin,3,3] -> fragments of the [9 Cout x Cin] matrix A[9 o + t, i] = wmul * W[o, i, t]: conv_transpose2d(x, W^T) before its
    scatter is A . x."""
    w = weight.detach()
    Cout, Cin = w.shape[:2]
    a = w.permute(0, 2, 3, 1).reshape(Cout * 9, Cin).contiguous()
    L = _lib.lib()
    blob = torch.empty(int(L.hav_gemm_packed_bytes(Cout * 9, Cin)), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(L.hav_gemm_pack(_p(blob), _p(a), Cout * 9, Cin, float(wmul), _stream(w.device)), "hav_gemm_pack")
    return blob


def upconv3x3(x, packed, Cout, fir, s=None, d=None, noise=None, noise_weight=None, bias=None, slope=0.2, gain=2 ** 0.5, act=True,
              autoscale=None):
    """y [B,Cout,2H,2W] = act(d * blur(conv_transpose2d(s * x, W, stride 2)) + noise_weight * noise + bias) * gain: the up-sampling
    StyledConv (model/styleUnet.py:236-243,565-599) as hav_gemm_split + hav_upconv_finish.  fir: the blur's [4,4] kernel (with the
    factor^2 gain folded in, as Blur holds it).  autoscale: as conv3x3 (range control of s * x before the fp16 split)."""
    if autoscale is None:
        autoscale = _autoscale_default()
    x = x.contiguous()
    B, Cin, H, W = x.shape
    col = torch.empty(B, Cout * 9, H * W, dtype=torch.float32, device=x.device)
    y = torch.empty(B, Cout, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
    nb = 0
    if noise is not None:
        noise = noise.contiguous()
        if noise.numel() == B * 4 * H * W and B > 1:
            nb = 1
        elif noise.numel() != 4 * H * W:
            raise RuntimeError("upconv3x3: noise must be [1,1,2H,2W] or [B,1,2H,2W]")
    fir = fir.contiguous()
    if tuple(fir.shape) != (4, 4) or fir.dtype != torch.float32:
        raise RuntimeError("upconv3x3: a [4,4] float32 FIR kernel is required")
    s, d, bias, noise_weight = _c(s), _c(d), _c(bias), _c(noise_weight)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        st = _stream(x.device)
        amax = _absmax(x, st) if autoscale else None
        _lib.check(L.hav_gemm_split(_p(col), _p(x), _p(packed), _p(s), _p(amax), B, Cout * 9, Cin, H * W, st), "hav_gemm_split")
        _lib.check(L.hav_upconv_finish(_p(y), _p(col), _p(fir), _p(d), _p(noise), _p(noise_weight), _p(bias), float(slope), float(gain),
                                       int(bool(act)), nb, B, Cout, H, W, st), "hav_upconv_finish")
    _trace("upconv3x3 x,col,y", x, col, y)
    return y


def upconv_block_eligible(x, weight):
    """x [B,Cin,H,W], weight [Cout,Cin,3,3]: shapes whose forward (hav_gemm_split + hav_upconv_finish) runs on this library's kernels; the
    data gradient does too where _upconv_dgrad_on_s2 says so."""
    return upconv_eligible(x, weight) and os.environ.get("HAVATAR_FUSED_UPBLOCK", "1") != "0"


def _upconv_dgrad_on_s2(x, weight):
    """can the data gradient of upconv_block(x, weight) run on hav_conv3x3s2_split?  (Its output is x-sized: 32-column tiles; the 16^2 -> 32^2
    layer of the encoders takes ATen's transposed-convolution gradient instead.)"""
    Cout, Cin = weight.shape[:2]
    B, _, H, W = x.shape
    return Cout % 16 == 0 and Cin % 64 == 0 and H % 4 == 0 and W % 32 == 0 and Cout * (2 * H + 1) * (2 * W + 1) < 2 ** 31


class _UpConvBlock(torch.autograd.Function):
    """y [B,Cout,2H,2W] = act(d * blur(conv_transpose2d(s * x, scale * W, stride 2)) + nw * noise + bias) * gain -- an up-sampling
    StyledConv (reference model/styleUnet.py:236-243,565-599) as ONE autograd node.  Forward = hav_gemm_split + hav_upconv_finish (the
    inference route).  Backward: hav_conv_block_bwd (activation gradient; d / bias / noise-weight gradients) -> the blur's adjoint
    (hav_upfirdn2d with the flipped FIR and padding (2, 2): [2H] -> [2H + 1]) -> the data gradient, which IS a stride-2 3x3 convolution of
    that map with the same filters (Cin and Cout swap roles; no flip: the adjoint of conv_transpose2d(., w) is conv2d(., w)) on
    hav_conv3x3s2_split -> hav_mod_input_bwd (s gradient, input scaling); the weight gradient is ATen's (MIOpen) on the same blurred map.
    Replaces ~25 ATen / MIOpen launches per layer and step (modulate, scale, transposed convolution + transposes, blur, demodulate,
    noise, bias, activation, and their gradients) by 9.  Under create_graph=True the backward restates the block with differentiable
    ATen ops; inside conv2d_gradfix.no_weight_gradients() no weight gradient is formed."""

    @staticmethod
    def forward(ctx, x, W, s, d, noise, nw, bias, fir, scale, slope, gain, act):
        y = upconv3x3(x, pack_upconv(W, scale), W.shape[0], fir, s=s, d=d, noise=noise, noise_weight=nw, bias=bias, slope=slope, gain=gain, act=act)
        ctx.save_for_backward(x, W, s, d, noise, nw, bias, fir, y)
        ctx.cfg = (float(scale), float(slope), float(gain), bool(act))
        return y

    @staticmethod
    def backward(ctx, g):
        from ..model.op import conv2d_gradfix
        from ..model.op.upfirdn2d import upfirdn2d as _ufd
        x, W, s, d, noise, nw, bias, fir, y = ctx.saved_tensors
        scale, slope, gain, act = ctx.cfg
        need = ctx.needs_input_grad
        B, Cout, H2, W2 = y.shape
        Cin, H, Wd = x.shape[1:]
        if torch.is_grad_enabled():          # create_graph=True: the block as differentiable ATen ops
            with torch.enable_grad():
                xs = x * s.view(B, Cin, 1, 1) if s is not None else x
                v = torch.nn.functional.conv_transpose2d(xs, (W * scale).transpose(0, 1), stride=2)
                v = torch.nn.functional.conv2d(torch.nn.functional.pad(v, (1, 1, 1, 1)), fir.flip(0, 1).view(1, 1, 4, 4).expand(Cout, 1, 4, 4), groups=Cout)
                if d is not None:
                    v = v * d.view(B, Cout, 1, 1)
                if noise is not None:
                    v = v + nw * noise
                if bias is not None:
                    v = v + bias.view(1, -1, 1, 1)
                if act:
                    v = torch.nn.functional.leaky_relu(v, slope) * gain
                slots = {0: x, 1: W, 2: s, 3: d, 5: nw, 6: bias}
                if conv2d_gradfix.weight_gradients_disabled:
                    slots.pop(1)
                idx = [k for k, t in slots.items() if t is not None and need[k]]
                got = torch.autograd.grad(v, [slots[k] for k in idx], g, create_graph=True, allow_unused=True) if idx else ()
            out = [None] * 12
            for k, gk in zip(idx, got):
                out[k] = gk
            return tuple(out)
        g = g.contiguous()
        L = _lib.lib()
        dev = y.device
        gc = torch.empty_like(y)
        sums = torch.empty(B * Cout * 3, dtype=torch.float32, device=dev)
        gd = torch.empty(B, Cout, dtype=torch.float32, device=dev) if (d is not None and need[3]) else None
        gb = torch.empty(Cout, dtype=torch.float32, device=dev) if (bias is not None and need[6]) else None
        gnw = torch.empty(1, dtype=torch.float32, device=dev) if (noise is not None and nw is not None and need[5]) else None
        nb = 1 if (noise is not None and B > 1 and noise.numel() == B * H2 * W2) else 0
        with torch.cuda.device(dev):
            _lib.check(L.hav_conv_block_bwd(_p(gc), _p(gd), _p(gb), _p(gnw), _p(sums), _p(g), _p(y), _p(d), _p(noise), _p(nw), _p(bias), slope, gain,
                                            int(act), nb, B, Cout, H2 * W2, _stream(dev)), "hav_conv_block_bwd")
        want_w = need[1] and not conv2d_gradfix.weight_gradients_disabled
        gx = gs = gW = None
        if need[0] or (s is not None and need[2]) or want_w:
            gv = _ufd(gc, fir.flip(0, 1), pad=(2, 2))          # d loss / d conv_transpose2d output  [B,Cout,2H+1,2W+1]
            wt = W.transpose(0, 1).contiguous()                          # [Cin,Cout,3,3]: conv_transpose2d's weight = the data gradient's conv weight
            if need[0] or (s is not None and need[2]):
                if _upconv_dgrad_on_s2(x, W):
                    gx = conv3x3s2(gv, pack(wt, scale), Cin, 0, act=False, autoscale=True)          # dL/d(s x): gradient-sized, see hav_absmax
                else:
                    gx, _, _ = torch.ops.aten.convolution_backward(gv, x, wt * scale, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1, [True, False, False])
                    gx = gx.contiguous()
                if s is not None:
                    gs = torch.empty(B, Cin, dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        _lib.check(L.hav_mod_input_bwd(_p(gx), _p(gs), _p(x), _p(s), B, Cin, H * Wd, _stream(dev)), "hav_mod_input_bwd")
            if want_w:
                if wgrad_s2_eligible(x, gv):
                    # d/dW of conv_transpose2d(s * x, W^T, stride 2) = the same contraction with x in the small-map role: one kernel, the
                    # modulation folded in, the parameter's [Cout,Cin,3,3] layout written directly
                    gW = wgrad3x3s2(x, gv, ss=s, out_mul=scale, transpose=True)
                else:
                    xs = x * s.view(B, Cin, 1, 1) if s is not None else x
                    _, gwt, _ = torch.ops.aten.convolution_backward(gv, xs, wt, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1, [False, True, False])
                    gW = gwt.transpose(0, 1) * scale
        if gnw is not None and nw.shape != gnw.shape:
            gnw = gnw.view(nw.shape)
        if gb is not None and bias.shape != gb.shape:
            gb = gb.view(bias.shape)
        _trace("UpConvBlock.bwd g,gc,gv,gx,gs,gW,gd,gnw,gb", g, gc, gv if (need[0] or want_w or (s is not None and need[2])) else None, gx, gs, gW, gd, gnw, gb)
        return (gx if need[0] else None), gW, (gs if need[2] else None), gd, None, gnw, gb, None, None, None, None, None


def upconv_block(x, W, scale, fir, s=None, d=None, noise=None, noise_weight=None, bias=None, slope=0.2, gain=2 ** 0.5, act=True):
    """see _UpConvBlock; x [B,Cin,H,W], W [Cout,Cin,3,3] raw parameter, fir the blur's [4,4] kernel (factor^2 folded in), s [B,Cin], d [B,Cout],
    noise [1|B,1,2H,2W] (not differentiated), noise_weight [1], bias [Cout]; callers check upconv_block_eligible(x, W)."""
    f = lambda t: None if t is None else t.contiguous()
    if noise is not None and noise_weight is None:
        noise = None
    return _UpConvBlock.apply(x.contiguous(), W.contiguous(), f(s), f(d), f(noise), f(noise_weight), f(bias), fir.detach().contiguous(), scale, slope,
                              gain, act)
