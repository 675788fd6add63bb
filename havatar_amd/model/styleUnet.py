"""StyleGAN2 / SWAGAN building blocks and the two networks the hot path calls: `StyleGAN_zxc` (tri-plane encoder, P3) and
`SWGAN_unet` (stage-two upsampler, P15).  Module code is plain PyTorch-ROCm (convs -> MIOpen); every blur / resample /
bias-activation goes through havatar_amd.model.op, i.e. the HIP kernels.

Behaviour and state_dict keys follow the reference (model/styleUnet.py:9-628 blocks, :631-878 StyleGAN_zxc,
:1190-1410 SWGAN_unet) so reference-trained checkpoints load; the implementation is written for this code base:
  * ModulatedConv2d uses the scale-input / shared-weight conv / scale-output factorisation (the reference's own
    `fused=False` algebra, :200-227) instead of materialising B*Cout per-sample filters for a grouped conv: one ordinary
    MIOpen convolution per call, identical mathematics (validated against the reference's fused branch).
  * only the configurations the path instantiates are supported (conditional-image encoder mode of StyleGAN_zxc).
"""
import math
import random

import torch
from torch import nn
from torch.nn import functional as F

import os

from .op import FusedLeakyReLU, conv2d_gradfix, fused_leaky_relu, upfirdn2d
from ..graph import weights_epoch
from ..native import conv as _conv


def _fused_conv_enabled():
    """HAVATAR_FUSED_CONV=0 keeps every convolution on MIOpen (A/B runs)."""
    return os.environ.get("HAVATAR_FUSED_CONV", "1") != "0"

_CHANNELS = lambda cm: {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}


class _InferenceCache:
    """Weight-only quantities (scale * W, sum_k W^2) are recomputed every call in the reference; at inference (no autograd)
    they are cached per module and keyed by the parameter's version, which removes ~70 elementwise launches over multi-MB
    weights from every frame.  Under autograd the cache is bypassed."""

    def _cached(self, name, param, fn):
        if torch.is_grad_enabled() and param.requires_grad:
            return fn()
        key = (param.data_ptr(), param._version, param.device, weights_epoch())
        c = self.__dict__.get("_icache")
        if c is None:
            c = self.__dict__["_icache"] = {}
        hit = c.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, fn())
            c[name] = hit
        return hit[1]


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        k = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", k)
        p = k.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        k = make_kernel(kernel)
        self.register_buffer("kernel", k)
        p = k.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer("kernel", k)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module, _InferenceCache):
    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        w = self._cached("w", self.weight, lambda: self.weight * self.scale)
        if (torch.is_grad_enabled() and _fused_conv_enabled() and input.shape[-1] * input.shape[-2] >= 256
                and _conv.eligible(input, w, self.stride, self.padding)):
            # training on HIP tensors: forward and both gradients on the split-fp16 MFMA kernels (native/conv.py)
            out = _conv.conv3x3_autograd(input, w)
            return out if self.bias is None else out + self.bias.view(1, -1, 1, 1)
        return conv2d_gradfix.conv2d(input, w, bias=self.bias, stride=self.stride, padding=self.padding)


class EqualLinear(nn.Module, _InferenceCache):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if (not self.activation and input.is_cuda and torch.is_grad_enabled() and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0"
                and os.environ.get("HAVATAR_EQUAL_LINEAR", "1") != "0"):
            from ..native.train_ops import equal_linear, equal_linear_eligible
            if equal_linear_eligible(input, self.weight, self.bias):
                # training: the scalar products over the parameters, the GEMM and their adjoints as one launch each way (hav_equal_linear_*)
                return equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul)
        w = self._cached("w", self.weight, lambda: self.weight * self.scale)
        b = self._cached("b", self.bias, lambda: self.bias * self.lr_mul) if self.bias is not None else None
        if self.activation:
            return fused_leaky_relu(F.linear(input, w), b)
        return F.linear(input, w, bias=b)


class ModulatedConv2d(nn.Module, _InferenceCache):
    """y = demod_b,o * conv(x * style_b,i , scale * W)   (style = EqualLinear(w), demod = rsqrt(sum (scale W style)^2 + eps))."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel=(1, 3, 3, 1), fused=True):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self.fused = fused

    def _hip_inference(self, input):
        return input.is_cuda and input.dtype == torch.float32 and not torch.is_grad_enabled()

    def fused_conv_ok(self, input):
        """3x3, no re-sampling, a shape hav_conv3x3_split takes (16^2 and up: 16-wide maps on the 8 x 16-tile variant; small maps run K-split over several workgroups per tile)."""
        return (self.kernel_size == 3 and not self.upsample and not self.downsample and _fused_conv_enabled()
                and input.shape[-1] * input.shape[-2] >= 256 and _conv.eligible(input, self.weight[0]))

    def packed3x3(self):
        return self._cached("w3x3", self.weight, lambda: _conv.pack(self.weight[0], self.scale))

    def fused_upconv_ok(self, input):
        """the up-sampling 3x3 (conv_transpose2d stride 2 -> 4x4 blur with padding (1,1)) in a shape hav_gemm_split + hav_upconv_finish take."""
        return (self.kernel_size == 3 and self.upsample and _fused_conv_enabled() and os.environ.get("HAVATAR_FUSED_UPCONV", "1") != "0"
                and tuple(self.blur.kernel.shape) == (4, 4) and tuple(self.blur.pad) == (1, 1) and _conv.upconv_eligible(input, self.weight[0]))

    def packed_upconv(self):
        return self._cached("wup", self.weight, lambda: _conv.pack_upconv(self.weight[0], self.scale))

    def style_vectors(self, style):
        """(s [B,Cin], d [B,Cout] | None).  HIP inference: one launch (hav_style_demod) instead of EqualLinear + bias + square +
        matmul + eps + rsqrt; otherwise the ATen sequence."""
        mod = self.modulation
        # sum_{i,ky,kx} (w[o,i] s[b,i])^2 = sum_i s[b,i]^2 * sum_k w[o,i,k]^2
        wsq_fn = lambda: (self._cached("wsq", self.weight, lambda: (self.scale * self.weight[0]).pow(2).sum((2, 3)).t().contiguous())
                          if self.demodulate else None)
        if self._hip_inference(style) and mod.activation is None:
            pre = self.__dict__.pop("_pre", None)          # computed up front with the generator's other layers (_prefetch_styles)
            if pre is not None:
                return pre
            from ..native import fused
            mw = mod._cached("w", mod.weight, lambda: mod.weight * mod.scale)
            mb = mod._cached("b", mod.bias, lambda: mod.bias * mod.lr_mul) if mod.bias is not None else None
            return fused.style_demod(style.contiguous(), mw, mb, wsq_fn(), self.eps)
        s = mod(style)
        if not self.demodulate:
            return s, None
        if s.is_cuda and s.dtype == torch.float32 and torch.is_grad_enabled() and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0":
            from ..native.train_ops import demod            # training on HIP tensors: one autograd node (hav_demod_fwd / _bwd)
            return s, demod(s, self.weight[0], self.scale, self.eps)
        return s, torch.rsqrt(torch.matmul(s * s, wsq_fn()) + self.eps)

    def forward_raw(self, input, style):
        """(conv(x * s) BEFORE demodulation, d): callers that fuse the demodulation into their epilogue use this."""
        B, Cin = input.shape[:2]
        w = self._cached("w", self.weight, lambda: self.scale * self.weight[0])          # [Cout,Cin,k,k]
        s, d = self.style_vectors(style)                                                  # [B,Cin], [B,Cout]
        x = input * s.view(B, Cin, 1, 1)
        if self.upsample:
            wt = self._cached("wt", self.weight, lambda: (self.scale * self.weight[0]).transpose(0, 1).contiguous())
            out = self.blur(conv2d_gradfix.conv_transpose2d(x, wt, padding=0, stride=2))
        elif self.downsample:
            out = conv2d_gradfix.conv2d(self.blur(x), w, padding=0, stride=2)
        elif (torch.is_grad_enabled() and self.kernel_size == 3 and _fused_conv_enabled() and x.shape[-1] * x.shape[-2] >= 256
              and _conv.eligible(x, w)):
            out = _conv.conv3x3_autograd(x, w)          # training: forward, data and weight gradient on the split-fp16 MFMA kernels (native/conv.py)
        else:
            out = conv2d_gradfix.conv2d(x, w, padding=self.padding)
        return out, d

    def forward(self, input, style):
        out, d = self.forward_raw(input, style)
        if d is not None:
            out = out * d.view(out.shape[0], -1, 1, 1)
        return out


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU]  (sequential indices are part of the checkpoint format)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=(1, 3, 3, 1), bias=True, activate=True):
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride, bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel, bias=bias))
        super().__init__(*layers)

    def forward(self, input):
        ec = self[0]
        if (len(self) > 1 and isinstance(self[0], Blur) and isinstance(self[1], EqualConv2d) and input.is_cuda and input.dtype == torch.float32
                and not torch.is_grad_enabled() and _fused_conv_enabled() and os.environ.get("HAVATAR_CONV_S2", "1") != "0"
                and os.environ.get("HAVATAR_S2_INFER", "1") != "0"):
            # HIP inference, down-sampling layer: Blur (hav_upfirdn2d) -> EqualConv2d stride 2 + bias + leaky-ReLU as one kernel
            # (hav_conv3x3s2_split) instead of MIOpen's Im2d2Col + fp32 GEMM + the activation launch: 78.6 against 110.8 us on the
            # encoders' 256 -> 512 layer, 49.5 against 69.1 us on the 512 -> 512 one, blur included (tools/bench_s2.py,
            # profiles/r04_bench_s2.txt; the first version of the kernel was slower than MIOpen and opt-in).  HAVATAR_CONV_S2=0: MIOpen
            ec = self[1]
            xb = self[0](input)
            if xb.shape[-1] * xb.shape[-2] >= 1024 and _conv.s2_eligible(xb, ec.weight, ec.stride, ec.padding):
                pk = ec._cached("w3x3", ec.weight, lambda: _conv.pack(ec.weight, ec.scale))
                if len(self) > 2:
                    return _conv.conv3x3s2(xb, pk, ec.weight.shape[0], ec.padding, bias=self[2].bias, slope=self[2].negative_slope,
                                           gain=self[2].scale, act=True)
                return _conv.conv3x3s2(xb, pk, ec.weight.shape[0], ec.padding, bias=ec.bias, act=False)
            out = ec(xb)
            return self[2](out) if len(self) > 2 else out
        if (len(self) > 1 and isinstance(self[0], Blur) and isinstance(self[1], EqualConv2d) and input.is_cuda and input.dtype == torch.float32
                and torch.is_grad_enabled() and _fused_conv_enabled() and os.environ.get("HAVATAR_CONV_S2", "1") != "0"
                and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0" and os.environ.get("HAVATAR_S2_TRAIN", "1") != "0"):
            # HIP training, down-sampling layer: the Blur (its own autograd op) -> EqualConv2d stride 2 + bias + leaky-ReLU as one autograd
            # node whose forward is hav_conv3x3s2_split (native/conv.py::_S2ConvBlock)
            ec, bl = self[1], self[0]
            p = bl.pad if len(bl.pad) == 4 else (bl.pad[0], bl.pad[1], bl.pad[0], bl.pad[1])
            kh, kw = bl.kernel.shape
            shape_b = (input.shape[0], input.shape[1], input.shape[2] + p[2] + p[3] - kh + 1, input.shape[3] + p[0] + p[1] - kw + 1)
            if shape_b[-1] * shape_b[-2] >= 1024 and _conv.s2_eligible(input, ec.weight, ec.stride, ec.padding, shape=shape_b):
                if len(self) > 2:
                    return _conv.s2_block(input, ec.weight, ec.scale, bias=self[2].bias, slope=self[2].negative_slope, gain=self[2].scale,
                                          act=True, padding=ec.padding, fir=bl.kernel, fir_pad=bl.pad)
                return _conv.s2_block(input, ec.weight, ec.scale, bias=ec.bias, act=False, padding=ec.padding, fir=bl.kernel, fir_pad=bl.pad)
            out = ec(bl(input))
            return self[2](out) if len(self) > 2 else out
        ec = self[0]
        if (isinstance(ec, EqualConv2d) and input.is_cuda and input.dtype == torch.float32 and torch.is_grad_enabled()
                and _fused_conv_enabled() and ec.stride == 1 and ec.padding == 1 and input.shape[-1] * input.shape[-2] >= 256
                and _conv.block_eligible(input, ec.weight)):
            # HIP training: EqualConv2d 3x3 + bias + leaky-ReLU as one autograd node (native/conv.py::_FusedConvBlock)
            if len(self) > 1:
                return _conv.fused_block(input, ec.weight, ec.scale, bias=self[1].bias, slope=self[1].negative_slope, gain=self[1].scale, act=True)
            return _conv.fused_block(input, ec.weight, ec.scale, bias=ec.bias, act=False)
        if (isinstance(ec, EqualConv2d) and input.is_cuda and input.dtype == torch.float32 and not torch.is_grad_enabled()
                and _fused_conv_enabled() and input.shape[-1] * input.shape[-2] >= 256
                and _conv.eligible(input, ec.weight, ec.stride, ec.padding)):
            # HIP inference: EqualConv2d 3x3 + bias + leaky-ReLU as one kernel (hav_conv3x3_split)
            act = len(self) > 1
            pk = ec._cached("w3x3", ec.weight, lambda: _conv.pack(ec.weight, ec.scale))
            if act:
                return _conv.conv3x3(input, pk, ec.weight.shape[0], bias=self[1].bias, slope=self[1].negative_slope, gain=self[1].scale, act=True)
            return _conv.conv3x3(input, pk, ec.weight.shape[0], bias=ec.bias, act=False)
        if (isinstance(ec, EqualConv2d) and input.is_cuda and input.dtype == torch.float32 and not torch.is_grad_enabled()
                and _fused_conv_enabled() and os.environ.get("HAVATAR_CONV_1X1", "0") == "1" and ec.stride == 1 and ec.padding == 0
                and _conv.conv1x1_eligible(input, ec.weight)):
            # HIP inference, on request (HAVATAR_CONV_1X1=1): the 1x1 EqualConv2d (FromRGB, conv_out) as a split-fp16 matrix product
            # (hav_gemm_split) instead of rocBLAS' fp32 GEMM.  Measured on SWGAN_unet 512 -> 1024 (K = 64, three layers): 2.74 against 2.70 ms
            # for the whole up-sampler -- with 2 k chunks the product is a pass over x and y, and the range-control pass in front of it costs
            # more than the fp32 GEMM's slower arithmetic: not the default
            pk = ec._cached("w1x1", ec.weight, lambda: _conv.pack_1x1(ec.weight, ec.scale))
            out = _conv.conv1x1(input, pk, ec.weight.shape[0])
            if len(self) > 1:
                return self[1](out)
            return out if ec.bias is None else out + ec.bias.view(1, -1, 1, 1)
        return super().forward(input)


def _fused_haar_enabled():
    return os.environ.get("HAVATAR_FUSED_HAAR", "1") != "0"


def _haar_bank(mod, ks):
    """[4,2,2] tensor of a transform's four kernels, cached on the module (buffers do not change after construction / .to())."""
    bank = mod.__dict__.get("_bank")
    if bank is None or bank.device != ks[0].device:
        bank = mod.__dict__["_bank"] = torch.stack([k.to(torch.float32) for k in ks]).contiguous()
    return bank


def get_haar_wavelet(in_channels):
    l = 1 / (2 ** 0.5) * torch.ones(1, 2)
    h = 1 / (2 ** 0.5) * torch.ones(1, 2)
    h[0, 0] = -h[0, 0]
    return l.T * l, h.T * l, l.T * h, h.T * h


class HaarTransform(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        for n, k in zip(("ll", "lh", "hl", "hh"), get_haar_wavelet(in_channels)):
            self.register_buffer(n, k)

    def forward(self, input):
        if input.is_cuda and input.dtype == torch.float32 and not torch.is_grad_enabled() and _fused_haar_enabled():
            from ..native import fused           # HIP inference: the four band filters + cat as one pass (hav_haar_dwt), same bits
            out = fused.haar(input, _haar_bank(self, (self.ll, self.lh, self.hl, self.hh)))
            if out is not None:
                return out
        return torch.cat([upfirdn2d(input, k, down=2) for k in (self.ll, self.lh, self.hl, self.hh)], 1)


class InverseHaarTransform(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        ll, lh, hl, hh = get_haar_wavelet(in_channels)
        for n, k in zip(("ll", "lh", "hl", "hh"), (ll, -lh, -hl, hh)):
            self.register_buffer(n, k)

    def forward(self, input):
        if input.is_cuda and input.dtype == torch.float32 and not torch.is_grad_enabled() and _fused_haar_enabled():
            from ..native import fused           # HIP inference: four up-sampling filters + three adds as one pass (hav_haar_idwt)
            out = fused.haar(input, _haar_bank(self, (self.ll, self.lh, self.hl, self.hh)), inverse=True)
            if out is not None:
                return out
        parts = input.chunk(4, 1)
        ks = (self.ll, self.lh, self.hl, self.hh)
        return sum(upfirdn2d(p, k, up=2, pad=(1, 0, 1, 0)) for p, k in zip(parts, ks))


class ConvBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1), downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample)

    def forward(self, input):
        return self.conv2(self.conv1(input))


class FromRGB(nn.Module):
    def __init__(self, out_channel, in_channel, downsample=True, blur_kernel=(1, 3, 3, 1), use_wt=True):
        super().__init__()
        self.downsample, self.use_wt = downsample, use_wt
        if downsample:
            self.downsample = Downsample(blur_kernel)
            if use_wt:
                self.iwt = InverseHaarTransform(in_channel)
                self.dwt = HaarTransform(in_channel)
        self.in_channel = in_channel * 4 if use_wt else in_channel
        self.conv = ConvLayer(self.in_channel, out_channel, 1)

    def forward(self, input, skip=None):
        if self.downsample:
            input = self.dwt(self.downsample(self.iwt(input))) if self.use_wt else self.downsample(input)
        out = self.conv(input)
        return input, (out if skip is None else out + skip)


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=(1, 3, 3, 1), demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample, blur_kernel=blur_kernel,
                                    demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        if self.conv._hip_inference(input) and self.conv.fused_conv_ok(input):
            # HIP inference, 3x3: modulation, the convolution (split-fp16 implicit GEMM on the matrix cores), demodulation, noise,
            # bias and leaky-ReLU as ONE kernel (hav_conv3x3_split)
            s, d = self.conv.style_vectors(style)
            if noise is None:
                b, _, h, w = input.shape
                noise = input.new_empty(b, 1, h, w).normal_()
            return _conv.conv3x3(input, self.conv.packed3x3(), self.conv.out_channel, s=s, d=d, noise=noise, noise_weight=self.noise.weight,
                                 bias=self.activate.bias, slope=self.activate.negative_slope, gain=self.activate.scale, act=True)
        if self.conv._hip_inference(input) and self.conv.fused_upconv_ok(input):
            # HIP inference, up-sampling 3x3: the transposed convolution as a split-fp16 matrix product, then scatter + blur + the
            # block's epilogue in one pass (hav_gemm_split + hav_upconv_finish) instead of MIOpen GEMM + Col2Im + upfirdn2d + epilogue
            s, d = self.conv.style_vectors(style)
            if noise is None:
                b, _, h, w = input.shape
                noise = input.new_empty(b, 1, 2 * h, 2 * w).normal_()
            return _conv.upconv3x3(input, self.conv.packed_upconv(), self.conv.out_channel, self.conv.blur.kernel, s=s, d=d, noise=noise,
                                   noise_weight=self.noise.weight, bias=self.activate.bias, slope=self.activate.negative_slope,
                                   gain=self.activate.scale, act=True)
        if (input.is_cuda and input.dtype == torch.float32 and torch.is_grad_enabled() and self.conv.kernel_size == 3
                and not self.conv.upsample and not self.conv.downsample and _fused_conv_enabled()
                and input.shape[-1] * input.shape[-2] >= 256 and _conv.block_eligible(input, self.conv.weight[0])):
            # HIP training, 3x3: the whole block as one autograd node (native/conv.py::_FusedConvBlock)
            s, d = self.conv.style_vectors(style)
            if noise is None:
                b, _, h, w = input.shape
                noise = input.new_empty(b, 1, h, w).normal_()
            return _conv.fused_block(input, self.conv.weight[0], self.conv.scale, s=s, d=d, noise=noise, noise_weight=self.noise.weight,
                                     bias=self.activate.bias, slope=self.activate.negative_slope, gain=self.activate.scale, act=True)
        if (input.is_cuda and input.dtype == torch.float32 and torch.is_grad_enabled() and self.conv.fused_upconv_ok(input)
                and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0" and _conv.upconv_block_eligible(input, self.conv.weight[0])):
            # HIP training, up-sampling 3x3: the whole block as one autograd node (native/conv.py::_UpConvBlock)
            s, d = self.conv.style_vectors(style)
            if noise is None:
                b, _, h, w = input.shape
                noise = input.new_empty(b, 1, 2 * h, 2 * w).normal_()
            return _conv.upconv_block(input, self.conv.weight[0], self.conv.scale, self.conv.blur.kernel, s=s, d=d, noise=noise,
                                      noise_weight=self.noise.weight, bias=self.activate.bias, slope=self.activate.negative_slope,
                                      gain=self.activate.scale, act=True)
        if self.conv._hip_inference(input):
            # HIP inference: demodulation * noise injection + bias + leaky-relu in ONE pass (hav_styled_epilogue) instead of four
            from ..native import fused
            out, d = self.conv.forward_raw(input, style)
            if noise is None:
                b, _, h, w = out.shape
                noise = out.new_empty(b, 1, h, w).normal_()
            return fused.styled_epilogue(out.contiguous(), d, noise.contiguous(), self.noise.weight, self.activate.bias,
                                         self.activate.negative_slope, self.activate.scale)
        return self.activate(self.noise(self.conv(input, style), noise=noise))


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, out_channel=12, upsample=True, blur_kernel=(1, 3, 3, 1), use_wt=True):
        super().__init__()
        self.use_wt = use_wt
        if upsample:
            self.upsample = Upsample(blur_kernel)
            if use_wt:
                self.iwt = InverseHaarTransform(3)
                self.dwt = HaarTransform(3)
        self.out_channel = out_channel if use_wt else out_channel // 4
        self.conv = ModulatedConv2d(in_channel, self.out_channel, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, self.out_channel, 1, 1))

    def forward(self, input, style, skip=None):
        if skip is not None:
            up = None
            if (self.use_wt and skip.is_cuda and skip.dtype == torch.float32 and not torch.is_grad_enabled() and _fused_haar_enabled()
                    and self.upsample.factor == 2 and tuple(self.upsample.pad) == (2, 1)):
                from ..native import fused           # HIP inference: synthesis, x2 FIR up-sampling and analysis as one pass (hav_haar_up2), same bits
                up = fused.haar_up2(skip, _haar_bank(self.iwt, (self.iwt.ll, self.iwt.lh, self.iwt.hl, self.iwt.hh)), self.upsample.kernel,
                                    _haar_bank(self.dwt, (self.dwt.ll, self.dwt.lh, self.dwt.hl, self.dwt.hh)))
            skip = up if up is not None else (self.dwt(self.upsample(self.iwt(skip))) if self.use_wt else self.upsample(skip))
        if self.conv._hip_inference(input) and os.environ.get("HAVATAR_FUSED_TORGB", "1") != "0":
            # HIP inference: modulation, the 1x1 convolution, bias and skip add in one pass over the activations (hav_torgb) instead of
            # x * s, MIOpen's GEMM between NHWC transposes, and two adds
            from ..native import fused
            s, _ = self.conv.style_vectors(style)
            out = fused.torgb(input, self.conv.weight[0], s, self.bias, skip, self.conv.scale)
            if out is not None:
                return out
        out = self.conv(input, style) + self.bias
        return out if skip is None else out + skip


def _prefetch_styles(owner, pairs, latent):
    """HIP inference: the (s, d) vectors of all modulated convolutions of a generator in ONE launch, before the first convolution
    (hav_style_demod_batched) -- they depend only on the latent.  pairs: [(ModulatedConv2d, index into latent[:, i])].  Each layer
    picks its pair up in style_vectors(); anything the batched kernel does not take falls back to the per-layer launch."""
    if not (latent.is_cuda and latent.dtype == torch.float32 and not torch.is_grad_enabled()
            and os.environ.get("HAVATAR_STYLE_BATCH", "1") != "0"):
        return
    from ..native import fused
    entries = []
    for mc, idx in pairs:
        mod = mc.modulation
        if mod.activation is not None or mc.eps != pairs[0][0].eps:
            return
        mw = mod._cached("w", mod.weight, lambda mod=mod: mod.weight * mod.scale)
        mb = mod._cached("b", mod.bias, lambda mod=mod: mod.bias * mod.lr_mul) if mod.bias is not None else None
        wsq = (mc._cached("wsq", mc.weight, lambda mc=mc: (mc.scale * mc.weight[0]).pow(2).sum((2, 3)).t().contiguous())
               if mc.demodulate else None)
        entries.append((mw, mb, wsq, idx))
    B = latent.shape[0]
    plan = owner.__dict__.get("_style_plan")
    if plan is None or plan.key != fused.StylePlan.make_key(entries, B, latent.device):
        plan = owner.__dict__["_style_plan"] = fused.StylePlan(entries, B, latent.device)
    for (mc, _), sd in zip(pairs, plan.run(latent.contiguous(), pairs[0][0].eps)):
        mc.__dict__["_pre"] = sd


def _mix_latents(styles, n_latent, inject_index):
    if len(styles) < 2:
        return styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    if inject_index is None:
        inject_index = random.randint(1, n_latent - 1)
    return torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                      styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)


def _style_mlp(in_dim, dim, n_mlp, lr_mlp):
    layers = [PixelNorm(), EqualLinear(in_dim, dim, lr_mul=lr_mlp, activation="fused_lrelu")]
    layers += [EqualLinear(dim, dim, lr_mul=lr_mlp, activation="fused_lrelu") for _ in range(n_mlp - 1)]
    return nn.Sequential(*layers)


def _run_style(style, z):
    """style(z) for a _style_mlp Sequential.  HIP inference with widths <= 64 (the tri-plane generators: 32 -> 32 x 4): PixelNorm and all
    layers as ONE launch (hav_style_mlp) instead of 5 + 2 n launch-bound ones at the head of the generator's chain; the blob of scaled,
    transposed weights is cached per weights epoch.  Anything else: the module itself."""
    if (z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and not torch.is_grad_enabled()
            and os.environ.get("HAVATAR_STYLE_MLP", "1") != "0" and isinstance(style, nn.Sequential) and len(style) >= 2
            and isinstance(style[0], PixelNorm)):
        layers = list(style)[1:]
        D = layers[0].weight.shape[0] if isinstance(layers[0], EqualLinear) else 0
        ok = (0 < D <= 64 and z.shape[1] <= 64 and all(isinstance(l, EqualLinear) and l.activation and l.bias is not None for l in layers)
              and layers[0].weight.shape[1] == z.shape[1] and all(tuple(l.weight.shape) == (D, D) for l in layers[1:]))
        if ok:
            from ..native import fused
            key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version) for l in layers) + (z.device, weights_epoch())
            hit = style.__dict__.get("_mlp_blob")
            if hit is None or hit[0] != key:
                with torch.no_grad():
                    parts = []
                    for l in layers:
                        parts += [(l.weight * l.scale).t().contiguous().reshape(-1), (l.bias * l.lr_mul).reshape(-1)]
                    hit = (key, torch.cat(parts).contiguous())
                style.__dict__["_mlp_blob"] = hit
            return fused.style_mlp(z, hit[1], len(layers), D)
    return style(z)


class _CondEncoder:
    """shared by both nets: conv_in (256->128) then FromRGB-pyramid + ConvBlocks, returning the feature list (fine -> coarse)."""

    def _encode(self, cond_img):
        out = self.conv_in(cond_img)
        feats = [out]
        for from_rgb, conv in zip(self.from_rgbs, self.cond_convs):
            cond_img, out = from_rgb(cond_img, out)
            out = conv(out)
            feats.append(out)
        return feats


class StyleGAN_zxc(nn.Module, _CondEncoder):
    """Tri-plane encoder: conditional image (inp_ch x inp_size^2) -> U-shaped StyleGAN2 synthesis at out_size.

    Only the conditional-image mode (inp_size > 0) is implemented -- the only one model/nerf_model.py instantiates."""

    def __init__(self, out_ch, out_size, style_dim, mlp_dim=32, n_mlp=0, middle_size=8, inject_layers=(), zero_latent=False,
                 zero_noise=False, no_skip=False, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01, n_latent=None,
                 inp_size=0, inp_ch=0, pass_kernel=False):
        super().__init__()
        if inp_size <= 0:
            raise NotImplementedError("StyleGAN_zxc: only the conditional-image encoder mode is on the hot path")
        self.no_skip = no_skip
        self.style_dim = mlp_dim
        self.middle_log_size = int(math.log(middle_size, 2))
        self.cond_img_enc = True
        self.n_mlp = n_mlp
        if n_mlp > 0:
            self.style = _style_mlp(style_dim, mlp_dim, n_mlp, lr_mlp)
        self.channels = ch = _CHANNELS(channel_multiplier)
        self.log_size = int(math.log(out_size, 2))

        in_channel = ch[inp_size // 2]
        self.from_rgbs, self.cond_convs, self.comb_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.comb_convs.append(ConvLayer(in_channel * 2, in_channel, 3))
        self.conv_in = ConvLayer(inp_ch, in_channel, 3, downsample=True)
        for i in range(int(math.log(inp_size, 2)) - 2, self.middle_log_size, -1):
            out_channel = ch[2 ** i]
            self.from_rgbs.append(FromRGB(in_channel, inp_ch, downsample=True, use_wt=False))
            self.cond_convs.append(ConvBlock(in_channel, out_channel, blur_kernel))
            self.comb_convs.append(ConvLayer(out_channel * 2, out_channel, 3))
            in_channel = out_channel

        self.convs, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.Module()
        self.input = ConstantInput(ch[middle_size], size=middle_size)
        self.conv1 = StyledConv(ch[middle_size], ch[middle_size], 3, self.style_dim, blur_kernel=blur_kernel)
        if no_skip:
            self.conv_out = ConvLayer(ch[out_size], out_ch, 1)
        else:
            self.to_rgb1 = ToRGB(ch[middle_size], self.style_dim, out_channel=out_ch * 4, upsample=False, use_wt=False)
        self.num_layers = (self.log_size - self.middle_log_size) * 2 + 1
        for li in range(self.num_layers):
            res = (li + 8) // 2
            self.noises.register_buffer(f"noise_{li}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = ch[middle_size]
        for i in range(self.middle_log_size + 1, self.log_size + 1):
            out_channel = ch[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, self.style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, self.style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(None if no_skip else ToRGB(out_channel, self.style_dim, out_channel=out_ch * 4, use_wt=False))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - (self.middle_log_size * 2 - 1) + 1 if n_latent is None else n_latent
        # zero_noise=True: all-zero injected noise EXCEPT the first entry, which is a random tensor drawn at construction,
        # kept as a plain attribute (not a buffer, not in the state_dict): reference :746-751 / SURVEY B-3.
        self.zero_noise = self.make_noise(zero_noise=True) if zero_noise else None
        if zero_latent:
            self.register_buffer("zero_latents", torch.zeros(1, self.n_latent, self.style_dim))
        else:
            self.zero_latents = None

    def make_noise(self, zero_noise=False):
        fn = torch.zeros if zero_noise else torch.randn
        noises = [torch.randn(1, 1, 2 ** self.middle_log_size, 2 ** self.middle_log_size)]
        for i in range(self.middle_log_size + 1, self.log_size + 1):
            noises += [fn(1, 1, 2 ** i, 2 ** i), fn(1, 1, 2 ** i, 2 ** i)]
        return noises

    def get_latent(self, styles, n_latent=None, inject_index=None):
        return _mix_latents([_run_style(self.style, s) for s in styles], self.n_latent if n_latent is None else n_latent, inject_index)

    def forward(self, styles, cond_feats, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, **kwargs):
        B = cond_feats.shape[0]
        if self.zero_latents is None:
            if not input_is_latent:
                assert self.n_mlp > 0
                styles = [_run_style(self.style, s) for s in styles]
            if truncation < 1:
                styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
            latent = _mix_latents(styles, self.n_latent, inject_index)
        else:
            latent = self.zero_latents.expand(B, -1, -1)
        if self.zero_noise is not None:
            noise = [n.to(cond_feats.device) for n in self.zero_noise]
            self.zero_noise = noise
        elif noise is None:
            noise = [None] * self.num_layers if randomize_noise else [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]

        pairs = [(self.conv1.conv, 0)] + ([] if self.no_skip else [(self.to_rgb1.conv, 1)])
        for k in range(len(self.convs) // 2):
            pairs += [(self.convs[2 * k].conv, 1 + 2 * k), (self.convs[2 * k + 1].conv, 2 + 2 * k)]
            if not self.no_skip:
                pairs.append((self.to_rgbs[k].conv, 3 + 2 * k))
        _prefetch_styles(self, pairs, latent)
        cond_list = self._encode(cond_feats)
        out = self.conv1(self.input(latent), latent[:, 0], noise=noise[0])
        skip = None if self.no_skip else self.to_rgb1(out, latent[:, 1])
        i = 1
        for conv_a, conv_b, n_a, n_b, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2], self.to_rgbs):
            if 1 < i <= 2 * len(cond_list) + 1:
                out = self.comb_convs[-(i // 2)](torch.cat([out, cond_list[-(i // 2)]], dim=1))
            out = conv_b(conv_a(out, latent[:, i], noise=n_a), latent[:, i + 1], noise=n_b)
            if not self.no_skip:
                skip = to_rgb(out, latent[:, i + 2], skip)
            i += 2
        image = self.conv_out(out) if self.no_skip else skip
        return (image, latent) if return_latents else (image, None)


class SWGAN_unet(nn.Module, _CondEncoder):
    """Stage-two upsampler: condition image [B,inp_ch,inp_size^2] -> RGB [B,3,out_size^2] in the Haar-wavelet domain."""

    def __init__(self, inp_size, inp_ch, out_ch, out_size, style_dim, n_mlp, middle_size=8, c_dim=0, channel_multiplier=2,
                 blur_kernel=(1, 3, 3, 1), lr_mlp=0.01):
        super().__init__()
        self.inp_size, self.style_dim = inp_size, style_dim
        self.middle_log_size = int(math.log(middle_size, 2))
        self.style = _style_mlp(style_dim + c_dim, style_dim, n_mlp, lr_mlp)
        self.channels = ch = _CHANNELS(channel_multiplier)
        self.log_size = int(math.log(out_size, 2)) - 1

        in_channel = ch[inp_size // 2]
        self.from_rgbs, self.cond_convs, self.comb_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.comb_convs.append(ConvLayer(in_channel * 2, in_channel, 3))
        self.conv_in = ConvLayer(inp_ch, in_channel, 3, downsample=True)
        for i in range(int(math.log(inp_size, 2)) - 2, self.middle_log_size - 1, -1):
            out_channel = ch[2 ** i]
            self.from_rgbs.append(FromRGB(in_channel, inp_ch, downsample=True, use_wt=False))
            self.cond_convs.append(ConvBlock(in_channel, out_channel, blur_kernel))
            self.comb_convs.append(ConvLayer(out_channel * 2 if i > self.middle_log_size else out_channel, out_channel, 3))
            in_channel = out_channel

        self.convs, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.Module()
        self.num_layers = (self.log_size - self.middle_log_size) * 2
        for li in range(self.num_layers):
            res = (li + 8) // 2
            self.noises.register_buffer(f"noise_{li}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = ch[middle_size]
        for i in range(self.middle_log_size + 1, self.log_size + 1):
            out_channel = ch[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim, out_channel=out_ch * 4))
            in_channel = out_channel
        self.iwt = InverseHaarTransform(3)
        self.n_latent = self.log_size * 2 - (self.middle_log_size * 2 - 1) + 1

    def make_noise(self, device, zero_noise=False):
        fn = torch.zeros if zero_noise else torch.randn
        return [fn(1, 1, 2 ** i, 2 ** i, device=device) for i in range(self.middle_log_size + 1, self.log_size + 1) for _ in range(2)]

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, condition_img, cond=None, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True):
        if not input_is_latent:
            styles = [_run_style(self.style, s if cond is None else torch.cat([s, cond], dim=-1)) for s in styles]          # (one launch on the device: hav_style_mlp)
        if noise is None:
            Hc, Wc = condition_img.shape[-2:]
            mid = 2 ** self.middle_log_size
            if (randomize_noise and condition_img.is_cuda and not torch.is_grad_enabled()
                    and (Hc * mid) % self.inp_size == 0 and (Wc * mid) % self.inp_size == 0):
                # inference on the device: every layer's fresh noise map from ONE normal_() launch instead of one per layer (the maps are
                # i.i.d. either way; 12 launches of ~5 us and their boundaries at 512 -> 1024).  Sizes follow the FEATURE MAPS, as
                # NoiseInjection's own `new_empty(b, 1, h, w)` does: the encoder takes an H x W condition image down to
                # (H, W) * middle_size / inp_size, decoder layer li works at that times 2^(li // 2 + 1) -- not the registered noise_i buffers,
                # whose sizes only match middle_size = 8 and a condition image of inp_size.  Anything that does not divide falls
                # back to per-layer noise.
                h0, w0 = Hc * mid // self.inp_size, Wc * mid // self.inp_size
                shapes = [(condition_img.shape[0], 1, h0 << (i // 2 + 1), w0 << (i // 2 + 1)) for i in range(self.num_layers)]
                sizes = [s[0] * s[2] * s[3] for s in shapes]
                flat = condition_img.new_empty(sum(sizes), dtype=torch.float32).normal_()
                noise, off = [], 0
                for s, n in zip(shapes, sizes):
                    noise.append(flat[off:off + n].view(s))
                    off += n
            else:
                noise = [None] * self.num_layers if randomize_noise else [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        latent = _mix_latents(styles, self.n_latent, inject_index)

        pairs = []
        for k in range(len(self.convs) // 2):
            pairs += [(self.convs[2 * k].conv, 2 * k), (self.convs[2 * k + 1].conv, 2 * k + 1), (self.to_rgbs[k].conv, 2 * k + 2)]
        _prefetch_styles(self, pairs, latent)
        cond_list = self._encode(condition_img)
        i, skip, out = 0, None, None
        for conv_a, conv_b, n_a, n_b, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[::2], noise[1::2], self.to_rgbs):
            if i == 0:
                out = self.comb_convs[-1](cond_list[-1])
            elif i < 2 * len(self.comb_convs):
                out = self.comb_convs[-1 - (i // 2)](torch.cat([out, cond_list[-1 - (i // 2)]], dim=1))
            out = conv_b(conv_a(out, latent[:, i], noise=n_a), latent[:, i + 1], noise=n_b)
            skip = to_rgb(out, latent[:, i + 2], skip)
            i += 2
        return self.iwt(skip)
