"""Tri-plane gather with gradients for the training path (hav_triplane_gather_fwd / _bwd): the HIP counterpart of
utils/util.py::sample_from_triplane_new under autograd.  HIP float32 tensors only."""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class TriplaneGather(Function):
    """feat [B*N, 2C] = gather(planes_cl [2,B,H,W,C], q [B,N,3]);  feat[:, 2c+p] = bilinear(plane p)[c]."""

    @staticmethod
    def forward(ctx, q, planes_cl):
        if not (q.is_cuda and planes_cl.is_cuda and q.dtype == torch.float32 and planes_cl.dtype == torch.float32):
            raise RuntimeError("TriplaneGather: HIP float32 tensors only")
        q = q.contiguous()
        planes_cl = planes_cl.contiguous()
        P, B, H, W, Cc = planes_cl.shape
        if P != 2 or q.shape[0] != B or q.shape[-1] != 3:
            raise RuntimeError("TriplaneGather: planes [2,B,H,W,C], q [B,N,3]")
        N = q.shape[1]
        feat = torch.empty(B * N, 2 * Cc, device=q.device, dtype=torch.float32)
        with torch.cuda.device(q.device):
            rc = _lib.lib().hav_triplane_gather_fwd(_p(feat), _p(planes_cl), _p(q), B * N, N, B, H, W, Cc, _stream())
        _lib.check(rc, "hav_triplane_gather_fwd")
        ctx.save_for_backward(q, planes_cl)
        return feat

    @staticmethod
    @once_differentiable
    def backward(ctx, dfeat):
        q, planes_cl = ctx.saved_tensors
        P, B, H, W, Cc = planes_cl.shape
        N = q.shape[1]
        dfeat = dfeat.contiguous()
        dplanes = torch.zeros_like(planes_cl)
        dq = torch.empty_like(q) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(q.device):
            rc = _lib.lib().hav_triplane_gather_bwd(_p(dplanes), _p(dq), _p(dfeat), _p(planes_cl), _p(q), B * N, N, B, H, W, Cc, _stream())
        _lib.check(rc, "hav_triplane_gather_bwd")
        return dq, (dplanes if ctx.needs_input_grad[1] else None)


def triplane_gather(q, planes_nchw):
    """q [B,N,3] box-warped coordinates, planes [2,B,C,H,W] (the Trainer's layout) -> [B*N, 2C] with gradients to both.
    The NCHW -> channels-last permutation is a differentiable ATen op, so the plane gradient arrives back in NCHW."""
    return TriplaneGather.apply(q, planes_nchw.permute(0, 1, 3, 4, 2).contiguous())
