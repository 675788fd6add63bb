"""GPU: the radiance MLP of the training path on the bf16 matrix cores (hav_mlp_train_*, BASELINE config 5) against
nn.Linear statements of model/nerf_model.py:104-117 -- fp32 autograd as the truth, a bf16-operand emulation as the yardstick."""
import numpy as np
import pytest
import torch

from havatar_amd import synth

pytestmark = pytest.mark.gpu
NAMES = ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")


def _weights(dev, scale=1.0, dtype=torch.float32):
    m = synth.scene(4, 4, "primary")["mlp"]
    return [torch.from_numpy(np.ascontiguousarray(m[k] * (scale if k.startswith("W") else 1.0))).to(dev, dtype).requires_grad_(True) for k in NAMES]


def _ref_forward(X, ws, bf16=False):
    """The reference statement; bf16=True rounds exactly the operands the kernel rounds (weights and the inputs of every matrix
    product), with everything else -- accumulation, biases, the alpha and rgb rows -- in the given dtype."""
    W1, b1, W2, b2, Wa, ba, Wf, bf, Wc, bc = ws
    r = (lambda t: t.to(torch.bfloat16).to(t.dtype)) if bf16 else (lambda t: t)
    h1 = torch.relu(r(X) @ r(W1).t() + b1)
    h2 = torch.relu(r(h1) @ r(W2).t() + b2)
    a = h2 @ Wa.t() + ba
    g = r(h2) @ r(Wf).t() + bf
    c = g @ Wc.t() + bc
    return torch.cat([c, g, a], -1)


def _inputs(n, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.cat([0.5 * torch.randn(n, 128, generator=g), torch.sin(8.0 * torch.randn(n, 48, generator=g))], 1)   # features | encoding
    d = torch.randn(n, 68, generator=g) / n
    return X.to(dev), d.to(dev)


@pytest.mark.parametrize("n", [32, 1000, 4096 + 17])
def test_forward_matches_reference_statement(n):
    from havatar_amd.native import mlp_train
    dev = torch.device("cuda:0")
    ws = _weights(dev)
    X, _ = _inputs(n, dev)
    rf = mlp_train.forward_only(X, mlp_train.pack(ws))
    ref64 = _ref_forward(X.double(), [w.double() for w in ws])
    emu = _ref_forward(X.double(), [w.double() for w in ws], bf16=True)
    scale = ref64.abs().max().item()
    assert rf.shape == (n, 68) and torch.isfinite(rf).all()
    assert (rf.double() - emu).abs().max().item() <= 1e-3 * scale        # same roundings; fp32-vs-fp64 accumulation flips a bf16 rounding here and there
    assert (rf.double() - ref64).abs().max().item() <= 2e-2 * scale      # bf16 operands: 2^-9 relative per operand
    assert (emu - ref64).abs().max().item() >= 0.2 * (rf.double() - ref64).abs().max().item()     # i.e. the error IS the bf16 rounding


@pytest.mark.parametrize("n", [64, 1000, 4096 + 17])
def test_backward_matches_autograd(n):
    """dX and all ten parameter gradients against fp64 autograd, with the error a bf16-operand emulation of the same statement makes
    as the yardstick: the kernel's gradients are bf16-class, not worse."""
    from havatar_amd.native import mlp_train
    dev = torch.device("cuda:0")
    ws = _weights(dev)
    X, d = _inputs(n, dev)
    Xg = X.clone().requires_grad_(True)
    out = mlp_train.fused_mlp(Xg, ws)
    out.backward(d)
    got = [Xg.grad] + [w.grad for w in ws]
    # fp64 truth and the bf16-operand emulation (autograd through the rounding as identity = what the kernel implements)
    def grads(bf16):
        w64 = [w.detach().double().requires_grad_(True) for w in ws]
        x64 = X.double().requires_grad_(True)
        if bf16:
            class R(torch.autograd.Function):
                @staticmethod
                def forward(ctx, t):
                    return t.to(torch.bfloat16).to(t.dtype)

                @staticmethod
                def backward(ctx, gout):
                    return gout
            W1, b1, W2, b2, Wa, ba, Wf, bf, Wc, bc = w64
            h1 = torch.relu(R.apply(x64) @ R.apply(W1).t() + b1)
            h2 = torch.relu(R.apply(h1) @ R.apply(W2).t() + b2)
            o = torch.cat([(R.apply(h2) @ R.apply(Wf).t() + bf) @ Wc.t() + bc, R.apply(h2) @ R.apply(Wf).t() + bf, h2 @ Wa.t() + ba], -1)
        else:
            o = _ref_forward(x64, w64)
        o.backward(d.double())
        return [x64.grad] + [w.grad for w in w64]
    ref, emu = grads(False), grads(True)
    for name, g, r, e in zip(("X",) + NAMES, got, ref, emu):
        scale = r.abs().max().item()
        err = (g.double() - r).abs().max().item()
        floor = (e - r).abs().max().item()
        assert g.shape == r.shape and torch.isfinite(g).all(), name
        # L-inf: not worse than the emulation.  bf16 operands move a pre-activation by ~1e-3 of the layer's scale, so a relu unit that
        # sits that close to zero switches on or off and its whole gradient term appears or vanishes: an O(1) error in the few
        # entries it touches, for the kernel and for the emulation alike (not necessarily the same units: hence the factor).
        out_layer = name in ("Wa", "ba", "Wf", "bf", "Wc", "bc")
        if not out_layer:
            assert err <= 3.0 * floor + 2e-3 * scale, (name, err / scale, floor / scale)
        # L2: the switches are rare (~1 % of the units), so in norm the gradients are 2e-2-class
        if n >= 1000:
            rel2 = ((g.double() - r).norm() / r.norm()).item()
            assert rel2 <= 8e-2, (name, rel2)
        if out_layer:          # no relu switch between them and the loss: plain bf16 rounding of their operands (the kernel also rounds
            assert err <= 2e-2 * scale, (name, err / scale)       # the upstream gradient, which the emulation does not)


def test_backward_is_bit_reproducible_and_accumulates_like_autograd():
    from havatar_amd.native import mlp_train
    dev = torch.device("cuda:0")
    ws = _weights(dev)
    X, d = _inputs(8192 + 5, dev, seed=3)
    runs = []
    for _ in range(3):
        for w in ws:
            w.grad = None
        Xg = X.clone().requires_grad_(True)
        mlp_train.fused_mlp(Xg, ws).backward(d)
        runs.append([Xg.grad.clone()] + [w.grad.clone() for w in ws])
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    for a, b in zip(runs[0], runs[2]):
        assert torch.equal(a, b)
    # two uses of the same parameters in one graph (the coarse and the fine pass of a step): autograd sums the two nodes' gradients
    for w in ws:
        w.grad = None
    (mlp_train.fused_mlp(X, ws).mul(d).sum() + mlp_train.fused_mlp(X, ws).mul(d).sum()).backward()
    for w, r in zip(ws, runs[0][1:]):
        assert torch.allclose(w.grad, 2.0 * r, rtol=0, atol=1e-6 * r.abs().max().item() + 1e-12)


def test_full_cfg5_pass_size_bias_gradients_are_column_sums():
    """At the size of cfg5's coarse pass (2 x 4096 rays x 64 samples = 524 288 queries): the gradients of the three output biases
    are plain sums of the upstream gradient over the queries -- a size-independent identity the slice-and-reduce path must keep
    (bf16 rounding of the summands is the only difference: d_c, d_a and dG enter the matrix cores as bf16)."""
    from havatar_amd.native import mlp_train
    dev = torch.device("cuda:0")
    ws = _weights(dev)
    n = 2 * 4096 * 64
    X, d = _inputs(n, dev, seed=5)
    d = d * n / 64.0
    Xg = X.clone().requires_grad_(True)
    mlp_train.fused_mlp(Xg, ws).backward(d)
    r = lambda t: t.to(torch.bfloat16).double()
    Wc = ws[8].detach().double()
    dg = d[:, 3:67].double() + d[:, :3].double() @ Wc
    for name, got, want in (("bc", ws[9].grad, r(d[:, :3]).sum(0)), ("ba", ws[5].grad, r(d[:, 67:]).sum(0)), ("bf", ws[7].grad, r(dg.float()).sum(0))):
        scale = want.abs().max().item() + d.abs().sum().item() / n * 1e-3
        assert (got.double() - want).abs().max().item() <= 2e-3 * max(scale, d.abs().max().item() * 30), name
    assert torch.isfinite(Xg.grad).all() and Xg.grad.abs().max().item() > 0


def test_field_inputs_and_mlp_as_one_node_with_bf16_rows_equal_the_two_nodes():
    """FieldMlp (native/train_ops.py: hav_field_inputs_fwd_bf16 -> hav_mlp_train_*_xbf16, the rows between the kernels in bf16) against
    FieldInputs -> FusedMlp (fp32 rows, rounded to bf16 inside the MLP kernels): the same rf bit for bit, the same MLP gradients bit for
    bit (fixed-order reduction), plane / volume gradients equal up to the order their float atomics land in."""
    from havatar_amd.native import mlp_train
    from havatar_amd.native.train_ops import field_inputs, field_mlp
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(31)
    nerf_box, skin_box = ([0.66, 0.65, 0.7], [0.0, 0.07, 0.14]), ([0.66, 1.9, 0.7], [0.0, -1.7, 0.14])
    B, R, S, Cc, H, D = 2, 512, 56, 64, 128, 64
    planes = torch.randn(2, B, Cc, H, H, device=dev, generator=g, requires_grad=True)
    vol0 = torch.sigmoid(2 * torch.randn(1, 1, D, D, D, device=dev, generator=g))
    vol = torch.cat([vol0, 1 - vol0], 1).requires_grad_(True)
    o = torch.rand(B, R, 1, 3, device=dev, generator=g) * 1.2 - 0.6
    d = torch.nn.functional.normalize(torch.randn(B, R, 1, 3, device=dev, generator=g), dim=-1)
    tt = torch.linspace(-1.2, 1.2, S, device=dev).view(1, 1, S, 1)
    pts = (o + d * tt).reshape(B, R * S, 3).contiguous()
    inv_T = torch.cat([torch.eye(3, device=dev).expand(B, 3, 3), torch.tensor([[[0.02, -0.03, 0.01]]], device=dev).expand(B, 1, 3)], 1).contiguous()
    ws = _weights(dev)
    up = torch.randn(B * R * S, 68, device=dev, generator=g) / (B * R * S)

    rf2 = mlp_train.fused_mlp(field_inputs(pts, inv_T, vol, planes, nerf_box, skin_box), ws)
    g2 = torch.autograd.grad(rf2, [planes, vol] + ws, up)
    rf1 = field_mlp(pts, inv_T, vol, planes, nerf_box, skin_box, ws)
    g1 = torch.autograd.grad(rf1, [planes, vol] + ws, up)
    assert torch.equal(rf1, rf2)
    for a, b, name in zip(g1[2:], g2[2:], NAMES):
        assert torch.equal(a, b), name
    for a, b, name in zip(g1[:2], g2[:2], ("planes", "vol")):
        scale = b.abs().max().item()
        assert scale > 0 and (a - b).abs().max().item() <= 2e-5 * scale, (name, (a - b).abs().max().item() / scale)
