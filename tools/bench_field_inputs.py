#!/usr/bin/env python3
"""hav_field_inputs_{fwd,bwd} at BASELINE config 5's size (2 frames x 4096 rays x 112 samples, ray-major queries, C = 64, 128^2 planes,
64^3 skinning volume): time per call of the run kernels (16 queries of a wave at a time: the forward's default, HAVATAR_FIELD_BWD=runs for the
backward) against the one-query-at-a-time kernels (HAVATAR_FIELD_BWD=walk; read once per process: each variant runs in a child process) -- X and both gradients must agree to fp32 rounding of the atomics' order."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from havatar_amd.native.train_ops import field_inputs
    DEV = torch.device("cuda:0")
    g = torch.Generator(device=DEV).manual_seed(23)
    nerf_box, skin_box = ([0.66, 0.65, 0.7], [0.0, 0.07, 0.14]), ([0.66, 1.9, 0.7], [0.0, -1.7, 0.14])
    B, R, S, Cc, H, D = 2, 4096, 112, 64, 128, 64
    planes = torch.randn(2, B, Cc, H, H, device=DEV, generator=g, requires_grad=True)
    vol0 = torch.sigmoid(2 * torch.randn(1, 1, D, D, D, device=DEV, generator=g))
    vol = torch.cat([vol0, 1 - vol0], 1).requires_grad_(True)
    o = torch.rand(B, R, 1, 3, device=DEV, generator=g) * 1.2 - 0.6
    d = torch.nn.functional.normalize(torch.randn(B, R, 1, 3, device=DEV, generator=g), dim=-1)
    tt = torch.linspace(-1.2, 1.2, S, device=DEV).view(1, 1, S, 1)
    pts = (o + d * tt).reshape(B, R * S, 3).contiguous()
    inv_T = torch.cat([torch.eye(3, device=DEV).expand(B, 3, 3), torch.tensor([[[0.02, -0.03, 0.01]]], device=DEV).expand(B, 1, 3)], 1).contiguous()
    up = torch.randn(B * R * S, 2 * Cc + 48, device=DEV, generator=g) * 1e-4

    def ev(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    X = field_inputs(pts, inv_T, vol, planes, nerf_box, skin_box)
    gp, gv = torch.autograd.grad(X, (planes, vol), up)
    with torch.no_grad():
        t_f = ev(lambda: field_inputs(pts, inv_T, vol, planes, nerf_box, skin_box))
    Xg = field_inputs(pts, inv_T, vol, planes, nerf_box, skin_box)
    t_b = ev(lambda: torch.autograd.grad(Xg, (planes, vol), up, retain_graph=True))
    np.savez(sys.argv[2], X=X.detach().cpu().numpy()[::97], gp=gp.cpu().numpy(), gv=gv.cpu().numpy())
    print(json.dumps({"mode": os.environ.get("HAVATAR_FIELD_BWD", "runs"), "queries": B * R * S, "fwd_ms": round(t_f, 3), "bwd_ms_incl_glue": round(t_b, 3)}))
    sys.exit(0)
import numpy as np, tempfile
tmp = tempfile.mkdtemp()
res = {}
for mode in ("walk", "runs"):
    env = dict(os.environ, HAVATAR_FIELD_BWD=mode)
    f = os.path.join(tmp, mode + ".npz")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
    res[mode] = np.load(f)
for k in ("X", "gp", "gv"):
    a, b = res["walk"][k], res["runs"][k]
    print("%-3s max |walk - runs| / max |walk| = %.3g   (max |walk| = %.3g)" % (k, np.abs(a - b).max() / np.abs(a).max(), np.abs(a).max()))
