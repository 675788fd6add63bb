"""fused bias + leaky-ReLU: Python surface of model/op/fused_act.py (reference :23-122), native part = HIP.

Autograd structure (first and second order) follows the op's contract:
  y  = s * lrelu_a(x + b)                                     native(act=3, grad=0)
  dx = s * (y > 0 ? g : a*g),  db = sum_{n,h,w} dx            native(act=3, grad=1, ref=y)
  d(dx)/dg applied to gg: same gate on gg + ggb               native(act=3, grad=1, ref=y, bias=ggb)
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

from ...native import fused


def _reduce_dims(t):
    return [0] + list(range(2, t.ndim))


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        none = grad_output.new_empty(0)
        grad_input = fused.fused_bias_act(grad_output.contiguous(), none, out, 3, 1, negative_slope, scale)
        grad_bias = grad_input.sum(_reduce_dims(grad_input)).detach() if bias else none
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        gg = fused.fused_bias_act(gradgrad_input.contiguous(), gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gg, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        none = input.new_empty(0)
        ctx.bias = bias is not None
        out = fused.fused_bias_act(input, none if bias is None else bias, none, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.bias, ctx.negative_slope, ctx.scale)
        return grad_input, (grad_bias if ctx.bias else None), None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type == "cpu":
        # The reference's CPU branch ignores `negative_slope` and hard-codes 0.2 (fused_act.py:113,119); kept as is.
        if bias is not None:
            input = input + bias.view(1, bias.shape[0], *([1] * (input.ndim - bias.ndim - 1)))
        return F.leaky_relu(input, negative_slope=0.2) * scale
    return FusedLeakyReLUFunction.apply(input.contiguous(), bias, negative_slope, scale)
