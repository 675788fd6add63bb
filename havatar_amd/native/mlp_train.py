"""The radiance MLP of the training path on the bf16 matrix cores (libhavatar_hip.so: hav_mlp_train_*), as one autograd node.

Replaces, for HIP tensors under autograd, the five nn.Linear calls of ConditionalTriplaneNeRFModel_multiRender_split_view.forward
(reference model/nerf_model.py:104-117).  Forward: X [n,176] -> rf [n,68].  Backward: activations are recomputed from X inside the
kernel (nothing but X and the packed weights is kept), dX and the ten parameter gradients come back in nn.Linear layouts.
Operands are rounded to bf16, accumulation and all tensors at this boundary are fp32 (BASELINE config 5: "bf16 MFMA MLP GEMM").
No fallback: a missing library raises."""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# (W1,b1,W2,b2,Wa,ba,Wf,bf,Wc,bc) of the radiance MLP the kernels are written for (reference model/nerf_model.py:77-117 with 2C + 48 = 176
# inputs, 128-wide hidden layers, a 64-wide feature layer and sh_deg 0: rgb 3)
_SHAPES = [(128, 176), (128,), (128, 128), (128,), (1, 128), (1,), (64, 128), (64,), (3, 64), (3,)]


def pack(weights):
    """Ten nn.Linear-layout fp32 tensors (W1,b1,W2,b2,Wa,ba,Wf,bf,Wc,bc) -> the bf16 fragment blob the kernels read."""
    L = _lib.lib()
    ws = [w.detach().contiguous() for w in weights]
    got = [tuple(w.shape) for w in ws]
    if got != _SHAPES or any(w.dtype != torch.float32 or not w.is_cuda for w in ws):
        # the kernels hard-code these sizes and take raw pointers: refuse anything else (rgb_feat_dim != 3, other hidden widths, sh_deg > 0)
        raise RuntimeError("mlp_train.pack: float32 HIP tensors of shapes %s required, got %s" % (_SHAPES, got))
    blob = torch.empty(int(L.hav_mlp_train_blob_bytes()), dtype=torch.uint8, device=ws[0].device)
    hw = _lib.HavMlpWeights(*[w.data_ptr() for w in ws])
    with torch.cuda.device(ws[0].device):
        _lib.check(L.hav_mlp_train_pack(_p(blob), C.byref(hw), _stream()), "hav_mlp_train_pack")
    return blob


def forward_only(X, blob):
    """X [n,176] float32, or bfloat16 rows as hav_field_inputs_fwd_bf16 writes them (same results: the kernel rounds fp32 rows to bf16 itself)"""
    n = X.shape[0]
    rf = torch.empty(n, 68, dtype=torch.float32, device=X.device)
    fn = _lib.lib().hav_mlp_train_fwd_xbf16 if X.dtype == torch.bfloat16 else _lib.lib().hav_mlp_train_fwd
    with torch.cuda.device(X.device):
        _lib.check(fn(_p(rf), _p(X), _p(blob), n, _stream()), "hav_mlp_train_fwd")
    return rf


def backward_only(X, d_rf, blob, shapes, need_dx=True):
    L = _lib.lib()
    n, dev = X.shape[0], X.device
    dX = torch.empty(X.shape, dtype=torch.float32, device=dev) if need_dx else None
    grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
    ops = torch.empty(int(L.hav_mlp_train_ops_bytes(n)), dtype=torch.uint8, device=dev)
    partial = torch.empty(int(L.hav_mlp_train_partial_bytes(n)), dtype=torch.uint8, device=dev)
    hg = _lib.HavMlpGrads(*[g.data_ptr() for g in grads])
    with torch.cuda.device(dev):
        fn = L.hav_mlp_train_bwd_xbf16 if X.dtype == torch.bfloat16 else L.hav_mlp_train_bwd
        _lib.check(fn(_p(dX), C.byref(hg), 0, _p(X), _p(d_rf), _p(blob), _p(ops), _p(partial), n, _stream()), "hav_mlp_train_bwd")
    return dX, grads


class FusedMlp(Function):
    @staticmethod
    def forward(ctx, X, *weights):
        if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.shape[1] == 176):
            raise RuntimeError("FusedMlp: X must be a float32 HIP tensor [n, 176]")
        X = X.contiguous()
        blob = pack(weights)
        ctx.save_for_backward(X, blob)
        ctx.shapes = [tuple(w.shape) for w in weights]
        return forward_only(X, blob)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_rf):
        X, blob = ctx.saved_tensors
        dX, grads = backward_only(X, d_rf.contiguous(), blob, ctx.shapes, need_dx=ctx.needs_input_grad[0])
        return (dX,) + tuple(grads)


def fused_mlp(X, weights):
    """weights = (W1,b1,W2,b2,Wa,ba,Wf,bf,Wc,bc) as returned by model.mlp_tensors()."""
    return FusedMlp.apply(X, *weights)


def bench_kernels(model, queries, dev, reps=5):
    """HIP-event time of the forward and backward kernels alone at the two pass sizes of a step (64/112 and 48/112 of `queries`)."""
    weights = [w.detach() for w in model.mlp_tensors()]
    blob = pack(weights)
    shapes = [tuple(w.shape) for w in weights]
    sizes = [queries * 64 // 112, queries * 48 // 112]
    data = [(torch.randn(n, 176, device=dev), torch.randn(n, 68, device=dev) * 1e-3) for n in sizes]

    def timed(fn):
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    for X, d in data:
        forward_only(X, blob); backward_only(X, d, blob, shapes)
    fwd = timed(lambda: [forward_only(X, blob) for X, _ in data])
    bwd = timed(lambda: [backward_only(X, d, blob, shapes) for X, d in data])
    n = sum(sizes)
    return {"fwd_ms": fwd, "bwd_ms": bwd, "mode": "bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16), recompute in backward",
            "bytes": n * 4 * (176 + 68) + n * 4 * (176 + 68 + 176)}
