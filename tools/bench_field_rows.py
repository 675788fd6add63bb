#!/usr/bin/env python3
"""hav_field_inputs_bwd against hav_field_inputs_bwd_rows on the rays of a training patch (a 64 x 64 block of neighbouring pixels of the bench's
pinhole camera, 112 samples per ray, 2 frames: BASELINE config 5's size): the same gradient sums, scattered 16 depths of one ray at a time
(ray-major runs) or 16 neighbouring rays of one depth at a time (rows).  Prints the time per call of both and the largest difference between
the gradients relative to the largest gradient (the float atomics land in another order: fp32 rounding)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from havatar_amd.native.train_ops import field_inputs
from havatar_amd import synth


def patch_queries(dev, B=2, S=112, side=64, gen=None):
    """pts [B, side^2 * S, 3]: the rays of rows/columns 224.. of a 512^2 view, S samples each between near and far, ray-major"""
    rays = torch.from_numpy(synth.camera_rays(512, 512)).float().to(dev).reshape(512, 512, -1)
    patch = rays[224:224 + side, 224:224 + side].reshape(-1, rays.shape[-1])
    o, d, near, far = patch[:, 0:3], patch[:, 3:6], patch[:, 6:7], patch[:, 7:8]
    t = torch.linspace(0, 1, S, device=dev)[None, :]
    z = near * (1 - t) + far * t
    if gen is not None:                                                   # stratified jitter, as the training pass draws it
        z = z + (torch.rand(z.shape, device=dev, generator=gen) - 0.5) * (far - near) / S
    return (o[:, None, :] + d[:, None, :] * z[:, :, None]).reshape(1, -1, 3).expand(B, -1, -1).contiguous()


def problem(dev, B=2, S=112, side=64, Cc=64, H=128, D=64, seed=23):
    g = torch.Generator(device=dev).manual_seed(seed)
    planes = torch.randn(2, B, Cc, H, H, device=dev, generator=g, requires_grad=True)
    vol0 = torch.sigmoid(2 * torch.randn(1, 1, D, D, D, device=dev, generator=g))
    vol = torch.cat([vol0, 1 - vol0], 1).requires_grad_(True)
    pts = patch_queries(dev, B, S, side, g)
    inv_T = torch.cat([torch.eye(3, device=dev).expand(B, 3, 3), torch.tensor([[[0.02, -0.03, 0.01]]], device=dev).expand(B, 1, 3)], 1).contiguous()
    up = torch.randn(B * side * side * S, 2 * Cc + 48, device=dev, generator=g) * 1e-4
    return planes, vol, pts, inv_T, up


BOXES = (([0.66, 0.65, 0.7], [0.0, 0.07, 0.14]), ([0.66, 1.9, 0.7], [0.0, -1.7, 0.14]))


def gradients(planes, vol, pts, inv_T, up, ray_rows):
    X = field_inputs(pts, inv_T, vol, planes, *BOXES, ray_rows=ray_rows)
    return X, torch.autograd.grad(X, (planes, vol), up, retain_graph=True)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    planes, vol, pts, inv_T, up = problem(dev)

    def ev(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2]
    out = {}
    for name, rows in (("ray_major", 0), ("rows", 112)):
        X, gr = gradients(planes, vol, pts, inv_T, up, rows)
        out[name] = {"bwd_ms_incl_glue": round(ev(lambda: torch.autograd.grad(X, (planes, vol), up, retain_graph=True)), 3)}
        out[name + "_g"] = gr
    for k, (a, b) in zip(("dplanes", "dvol"), zip(out.pop("ray_major_g"), out.pop("rows_g"))):
        out["rel_diff_" + k] = float((a - b).abs().max() / a.abs().max())
    print(json.dumps(out))
