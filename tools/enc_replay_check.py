#!/usr/bin/env python3
"""Replay determinism of the tri-plane encoders inside a hipGraph: set_conditional_embedding captured with its two generators on two
streams (the default) or on one (HAVATAR_ENC_STREAMS=0); REPLAYS replays, each compared bitwise with the first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import synth
from havatar_amd.graph import GraphedForward
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

REPLAYS = int(os.environ.get("REPLAYS", "300"))
dev = torch.device("cuda:0")
cfg = CfgNode(synth.harness_config(render_size=128, gen_size=512, img_res=512, perturb=False, noise_std=0.0, rays=4096))
torch.manual_seed(0)
tr = synth.fill_state_dict(Trainer(cfg, 2)).to(dev).eval()
m = tr.model_coarse
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
front, left, right = [t(a) for a in synth.cond_images()]
inv_T = t(synth.frame_pose(0))[None]
lat = tr.latent_codes[0:1] if tr.latent_codes is not None else None


class Enc(torch.nn.Module):
    def forward(self, front_render_cond, left_render_cond, right_render_cond, cond_c):
        m.set_conditional_embedding(front_render_cond=front_render_cond, left_render_cond=left_render_cond, right_render_cond=right_render_cond,
                                    latents=lat, cond_c=cond_c)
        return m.triPlane_embeddings


data = dict(front_render_cond=front, left_render_cond=left, right_render_cond=right, cond_c=inv_T.reshape(1, -1))
with torch.no_grad():
    eager = Enc()(**data).clone()
g = GraphedForward(Enc(), data)
first = g(**data).clone()
torch.cuda.synchronize()
differ, worst = 0, 0.0
for i in range(REPLAYS):
    out = g(**data)
    torch.cuda.synchronize()
    if not torch.equal(out, first):
        differ += 1
        worst = max(worst, (out - first).abs().max().item())
print("HAVATAR_ENC_STREAMS=%s: %d of %d replays differ from the first (worst %.3e); first replay vs eager: max diff %.3e, finite %s" % (
    os.environ.get("HAVATAR_ENC_STREAMS", "1"), differ, REPLAYS, worst, (first - eager).abs().max().item(), bool(torch.isfinite(first).all())))
