#!/usr/bin/env python3
"""Feasibility of pipelined frames (BASELINE config 3, a SEQUENCE of frames): the march of frame i is a persistent kernel that owns every compute
unit it runs on (2 waves per SIMD at 256 VGPRs, 152 KB of LDS), the encoders of frame i + 1 are ~150 small launches that leave most of the chip
idle.  Run them concurrently: the march on G < 256 compute units (HavRenderParams.grid_blocks), the encoders + plane preparation of the next
frame on a second stream, which the dispatcher places on the units the march leaves free.  Prints ms per frame for G in GRIDS against the
sequential frame (encoders, then the march on all units)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import synth
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
H = W = 512
cfg = CfgNode.load_yaml(os.path.join(ROOT, "havatar_amd", "config", "hd_base.yml"))
cfg.models.StyleUnet.inp_size = H
v = cfg.nerf.validation
v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = 64, 16, True, 0.0
torch.manual_seed(0)
tr = Trainer(cfg, 1)
tr.requires_grad_(False)
synth.fill_state_dict(tr)
tr = tr.to(dev).eval()
tr.headpose_skin_net.fix_canonical_W()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
front, left, right = [t(a) for a in synth.cond_images()]
rays = t(synth.camera_rays(H, W))[None]
rd = rays[..., 3:6]
rays11 = torch.cat((rays, rd / rd.norm(p=2, dim=-1).unsqueeze(-1)), dim=-1)
bg = torch.ones(1, H * W, 3, device=dev)
pose = t(synth.frame_pose(0))[None]
m = tr._hip_marcher()
vol = tr.headpose_skin_net.current_volume().detach()
N = int(os.environ.get("FRAMES", "60"))


def encoders():
    tr.model_coarse.set_conditional_embedding(front_render_cond=front, left_render_cond=left, right_render_cond=right,
                                              latents=tr.latent_codes[0:1], cond_c=pose.view(1, -1))
    m.set_triplane(tr.model_coarse.triPlane_embeddings)


def march():
    return m.render(rays11, bg, pose, vol, 64, 16, perturb=True, noise_std=0.0, coarse_outputs=False)


def capture(fn):
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            fn()
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = fn()
    return g, out


with torch.no_grad():
    m.set_mlp(*[x.detach() for x in tr.model_coarse.mlp_tensors()])
    encoders()
    march()
torch.cuda.synchronize()
g_enc, _ = capture(encoders)
s_main, s_side, s_fork = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def timeit(fn, n=N):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("encoders alone (one graph, two internal streams): %.3f ms" % timeit(g_enc.replay))
for G in [int(x) for x in os.environ.get("GRIDS", "256,232,208,176,152").split(",")]:
    m.grid_blocks = 0 if G >= 256 else G
    g_march, _ = capture(march)
    t_march = timeit(g_march.replay)

    def sequential():
        g_enc.replay()
        g_march.replay()

    def pipelined():
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s_side):
            s_side.wait_event(ev)
            g_enc.replay()
        with torch.cuda.stream(s_main):
            s_main.wait_event(ev)
            g_march.replay()
        torch.cuda.current_stream(dev).wait_stream(s_side)
        torch.cuda.current_stream(dev).wait_stream(s_main)

    def forked():            # ONE graph: the march on the capture stream, the encoders on a forked branch
        side = s_fork
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            os.environ["HAVATAR_ENC_STREAMS"] = os.environ.get("FORK_NESTED", "0")          # 0: both generators on this one branch (no nested fork)
            encoders()
            os.environ["HAVATAR_ENC_STREAMS"] = "1"
        out = march()
        cur.wait_stream(side)
        return out
    g_fork, _ = capture(forked)
    print("march on %3d compute units: alone %.3f ms | encoders then march %.3f ms per frame | two graphs on two streams %.3f | ONE graph, encoders of the next frame on a forked branch beside the march %.3f ms per frame (%s)" % (
        G, t_march, timeit(sequential), timeit(pipelined), timeit(g_fork.replay), m.last_variant), flush=True)
