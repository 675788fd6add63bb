"""upfirdn2d: Python surface of model/op/upfirdn2d.py (reference :22-213), native part = HIP.

Gradient identity (SURVEY A-9): dL/dx = upfirdn2d(dL/dy, flip(k), up=down, down=up, pad=g_pad) with
g_pad = (kw-px0-1, W*up-out_w*down+px0-up+1, kh-py0-1, H*up-out_h*down+py0-up+1); the second-order term is the
forward op again.
"""
from collections import abc

import torch
from torch.autograd import Function
from torch.nn import functional as F

from ...native import upfirdn2d as upfirdn2d_op


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        g = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        if not g.is_contiguous():
            g = g.contiguous()
        gi = upfirdn2d_op.upfirdn2d(g, grad_kernel, down[0], down[1], up[0], up[1], *g_pad)
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.in_size, ctx.out_size = in_size, out_size
        return gi.view(in_size[0], in_size[1], in_size[2], in_size[3])

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        gg = gradgrad_input.reshape(-1, ctx.in_size[2], ctx.in_size[3], 1)
        if not gg.is_contiguous():
            gg = gg.contiguous()
        out = upfirdn2d_op.upfirdn2d(gg, kernel, ctx.up[0], ctx.up[1], ctx.down[0], ctx.down[1], *ctx.pad)
        out = out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
        return out, None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        x = input.reshape(-1, in_h, in_w, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
        out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (px0, px1, py0, py1)
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        out = upfirdn2d_op.upfirdn2d(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down, ctx.pad,
                                                 ctx.g_pad, ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """input [B,C,H,W], kernel [kh,kw]; up/down int or (x,y); pad (p0,p1) or (x0,x1,y0,y1)."""
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    if input.device.type == "cpu":
        return upfirdn2d_native(input, kernel, *up, *down, *pad)
    if not (torch.is_grad_enabled() and input.requires_grad):
        # inference: straight to the native op (the autograd Function would flip the FIR for a backward that never comes:
        # one more launch per call, ~50 per frame)
        _, channel, in_h, in_w = input.shape
        x = input.reshape(-1, in_h, in_w, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        out = upfirdn2d_op.upfirdn2d(x, kernel, up[0], up[1], down[0], down[1], *pad)
        return out.view(-1, channel, out.shape[1], out.shape[2])
    return UpFirDn2d.apply(input, kernel, tuple(up), tuple(down), tuple(pad))


def upfirdn2d_native(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Pure-PyTorch path for CPU tensors (role of reference :172-213): zero-stuff, pad/crop, conv with the flipped
    FIR, decimate."""
    _, channel, in_h, in_w = input.shape
    kh, kw = kernel.shape
    x = input.reshape(-1, 1, in_h, in_w)
    if up_x > 1 or up_y > 1:
        z = x.new_zeros(x.shape[0], 1, in_h * up_y, in_w * up_x)
        z[:, :, ::up_y, ::up_x] = x
        x = z
    x = F.pad(x, [max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    x = x[:, :, max(-pad_y0, 0): x.shape[2] - max(-pad_y1, 0), max(-pad_x0, 0): x.shape[3] - max(-pad_x1, 0)]
    y = F.conv2d(x, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(x.dtype))
    y = y[:, :, ::down_y, ::down_x]
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) // down_y
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) // down_x
    return y.reshape(-1, channel, out_h, out_w)
