"""`upfirdn2d` -- same call surface as the reference's pybind module (model/op/upfirdn2d.cpp:17-31),
backed by hav_upfirdn2d in libhavatar_hip.so."""
import ctypes as C

import torch

from .. import _lib

_DT = {torch.float32: _lib.HAV_F32, torch.float16: _lib.HAV_F16, torch.bfloat16: _lib.HAV_BF16,
       torch.float64: _lib.HAV_F64}


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """input [major,in_h,in_w,minor], kernel [kh,kw] -> [major,out_h,out_w,minor]."""
    for t, n in ((input, "input"), (kernel, "kernel")):     # CHECK_INPUT (upfirdn2d.cpp:10-16,21-22)
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{n} must be contiguous")
    if input.dim() != 4 or kernel.dim() != 2:
        raise RuntimeError("upfirdn2d: input must be [major,H,W,minor] and kernel [kh,kw]")
    if input.dtype not in _DT:
        raise RuntimeError(f"upfirdn2d: unsupported dtype {input.dtype}")
    major, in_h, in_w, minor = input.shape
    kh, kw = kernel.shape
    k = kernel.to(torch.float32).contiguous()
    L = _lib.lib()
    oh, ow = C.c_int(0), C.c_int(0)
    args = [int(v) for v in (up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)]
    _lib.check(L.hav_upfirdn2d_out_size(in_h, in_w, kh, kw, *args, C.byref(oh), C.byref(ow)), "hav_upfirdn2d_out_size")
    out = torch.empty((major, oh.value, ow.value, minor), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(input.device):
        rc = L.hav_upfirdn2d(C.c_void_p(out.data_ptr()), C.c_void_p(input.data_ptr()), C.c_void_p(k.data_ptr()),
                             _DT[input.dtype], major, in_h, in_w, minor, kh, kw, *args,
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "hav_upfirdn2d")
    return out
