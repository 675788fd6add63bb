#!/usr/bin/env python3
"""Batched frames (BASELINE config 3, throughput mode): B frames per hipGraph replay -- the encoders see a batch (their 16^2 .. 64^2 layers cannot
fill 256 compute units at B = 1), the march takes B x R rays in one launch.  Prints ms per frame of the encoders alone and of the whole frame for
B in BATCHES, and how far the frames of a batch are from the same frames rendered one at a time (deterministic depths)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import synth
from havatar_amd.graph import GraphedForward
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
H = W = int(os.environ.get("SIZE", "512"))
PERTURB = os.environ.get("PERTURB", "1") == "1"
cfg = CfgNode.load_yaml(os.path.join(ROOT, "havatar_amd", "config", "hd_base.yml"))
cfg.models.StyleUnet.inp_size = 512
v = cfg.nerf.validation
v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = 64, 16, PERTURB, 0.0
torch.manual_seed(0)
tr = Trainer(cfg, 1)
tr.requires_grad_(False)
synth.fill_state_dict(tr)
tr = tr.to(dev).eval()
tr.headpose_skin_net.fix_canonical_W()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
front, left, right = [t(a) for a in synth.cond_images()]
rays = t(synth.camera_rays(H, W))[None]
N = int(os.environ.get("FRAMES", "40"))


def batch(B, first=0):
    poses = torch.cat([t(synth.frame_pose(first + k))[None] for k in range(B)])
    rep = lambda x: x.expand(B, *x.shape[1:]).contiguous()
    # every frame of the batch gets its own condition images (here: the synthetic ones, scaled per frame so that the planes differ)
    sc = torch.tensor([1.0 - 0.05 * ((first + k) % 4) for k in range(B)], device=dev).view(B, 1, 1, 1)
    return dict(ray_batch=rep(rays), background_prior=torch.ones(B, H * W, 3, device=dev), inv_head_T=poses, front_render_cond=rep(front) * sc,
                left_render_cond=rep(left) * sc, right_render_cond=rep(right) * sc, mode="validation", fidx=0, render_full_img=True)


def timeit(fn, n):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


ref = {}
for B in [int(x) for x in os.environ.get("BATCHES", "1,2,4").split(",")]:
    data = batch(B)
    g = GraphedForward(tr, data)
    ms = timeit(lambda: g(inv_head_T=data["inv_head_T"]), max(4, N // B))

    def enc():
        tr.model_coarse.set_conditional_embedding(front_render_cond=data["front_render_cond"], left_render_cond=data["left_render_cond"],
                                                  right_render_cond=data["right_render_cond"], latents=tr.latent_codes[0:1].expand(B, -1),
                                                  cond_c=data["inv_head_T"].reshape(B, -1))
    ge = GraphedForward(lambda inv_head_T: enc(), {"inv_head_T": data["inv_head_T"]})
    ms_e = timeit(lambda: ge(inv_head_T=data["inv_head_T"]), max(4, N // B))
    out = g(inv_head_T=data["inv_head_T"])[0].clone()
    line = "B = %d: frame %.3f ms per frame (%.1f frames/s), encoders %.3f ms per frame" % (B, ms / B, 1e3 * B / ms, ms_e / B)
    if not PERTURB:
        if B == 1:
            for k in range(4):
                d1 = batch(1, first=k)
                ref[k] = GraphedForward(tr, d1)(inv_head_T=d1["inv_head_T"])[0].clone()
        else:
            err = max(float((out[k:k + 1] - ref[k]).abs().max()) for k in range(B))
            line += " | frames of the batch vs rendered one at a time: L-inf %.2e" % err
    print(line, flush=True)
    del g, ge
