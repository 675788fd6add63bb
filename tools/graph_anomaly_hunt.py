#!/usr/bin/env python3
"""How often does a replay of the graphed optimisation step flag a non-finite tensor?

In ONE process: BUILDS times a fresh Trainer + optimiser + StepRunner (2 eager steps, capture), then STEPS replays each with an eager
inference render every RENDER_EVERY steps (what train.main() does between steps, and where round 4 saw its two events: the second replay
after a render).  After every replay the is-finite flags that native/conv.py::_trace evaluates INSIDE the graph are read back.

HAVATAR_NAN_TRACE selects the tracer: 1 = torch.isfinite(t).all() (temporaries in the graph's memory pool), 2 = hav_debug_nonfinite into a
buffer outside the pool (no temporaries).  SENTINEL=1 additionally pre-fills every hav_absmax word buffer with NaN bit patterns inside the
graph, so a consumer that runs before its absmax launch has finished shows up as a flagged `amax` operand.
Output: one line per flagged replay (build, step, steps since the last render, names) and a summary line."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import synth
from havatar_amd.dataloader.dataloader import Loader
from havatar_amd.harness import train
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.native import conv
from havatar_amd.utils.cfgnode import CfgNode

GRAPH = os.environ.get("GRAPH", "1") != "0"          # GRAPH=0: the same loop with eager steps (is an event a property of the replay or of a kernel?)
BUILDS, STEPS, EVERY = int(os.environ.get("BUILDS", "3")), int(os.environ.get("STEPS", "200")), int(os.environ.get("RENDER_EVERY", "4"))
assert conv._NAN_TRACE is not None, "set HAVATAR_NAN_TRACE=1|2"
if os.environ.get("SENTINEL") == "1":
    _orig = conv._absmax

    def _absmax_sentinel(t, st):
        words = torch.empty(256, dtype=torch.int32, device=t.device)
        words.fill_(-1)                                  # 0xFFFFFFFF: a NaN pattern, larger than any float's bits in the unsigned fold
        conv._lib.check(conv._lib.lib().hav_absmax(conv._p(words), conv._p(t), t.numel(), st), "hav_absmax")
        return words
    conv._absmax = _absmax_sentinel

KIND = os.environ.get("RENDER_KIND", "full")          # what runs between replays: full | encoders | toggle | alloc
PROBE = os.environ.get("PROBE") == "1"          # PROBE=1: which blocks of the graph's PRIVATE memory pool change hands around an eager render?


def pool_blocks():
    """{address: (size, state)} of every block in a segment that belongs to a private (graph) pool"""
    out = {}
    for seg in torch.cuda.memory_snapshot():
        if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
            for blk in seg["blocks"]:
                out[blk["address"]] = (blk["size"], blk["state"])
    return out


def pool_diff(tag, a, b):
    gone = [k for k in a if k not in b]
    new = [k for k in b if k not in a]
    chg = [(k, a[k], b[k]) for k in a if k in b and a[k] != b[k]]
    act = sum(1 for v in b.values() if v[1] == "active_allocated")
    print("pool probe %s: %d blocks (%d allocated, %.1f MB); vs before: %d gone, %d new, %d changed state" % (
        tag, len(b), act, sum(v[0] for v in b.values()) / 2 ** 20, len(gone), len(new), len(chg)), flush=True)
    for k, x, y in chg[:12]:
        print("     0x%x  %d B  %s -> %s" % (k, x[0], x[1], y[1]), flush=True)
    for k in new[:6]:
        print("     new 0x%x  %d B  %s" % (k, b[k][0], b[k][1]), flush=True)


dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
split = synth.write_dataset(tmp, n_frames=2, img_res=128)
cfg = CfgNode(synth.harness_config(perturb=True, noise_std=0.1))
np.random.seed(3)
tl = Loader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg, white_bg=True, shuffle=False)
idx, batch = next(iter(tl))
events, replays, t0, snap = [], 0, time.time(), {}
for b in range(BUILDS):
    torch.manual_seed(11 + b)
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(dev).train()
    opt = train.make_optimizer(cfg, trainer, True)
    run = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=GRAPH)
    inp, target, mask = train.step_inputs(idx, batch, dev)
    since = -1
    for k in range(STEPS + 2):
        if not GRAPH:
            conv._NAN_TRACE.clear()
            conv.nan_trace_reset(dev)
        loss, parts, _ = run(inp, target, mask)
        train.set_learning_rate(opt, 5e-4)
        if PROBE and run.graphed is not None and k <= 8:
            torch.cuda.synchronize()
            now = pool_blocks()
            pool_diff("after step %d (replay)" % k, snap if k > 2 else now, now)
            snap = now
        if run.graphed is not None or (not GRAPH and k >= 2):
            replays += 1
            since = since + 1 if since >= 0 else -1
            bad = [n for n, f in list(conv._NAN_TRACE) if not bool(f)]
            if bad or not np.isfinite(loss.item()):
                events.append((b, k, since, loss.item(), bad[:6]))
                nanparts = [n for n, v in parts.items() if torch.is_tensor(v) and not bool(torch.isfinite(v).all())]
                wbad = sum(int(not bool(torch.isfinite(p_).all())) for p_ in trainer.parameters())
                gbad = [n_ for n_, p_ in trainer.named_parameters() if p_.grad is not None and not bool(torch.isfinite(p_.grad).all())]
                pbad = [n_ for n_, p_ in trainer.named_parameters() if not bool(torch.isfinite(p_).all())]
                print("   non-finite gradients: %d %s | non-finite parameters: %s" % (len(gbad), gbad[:6], pbad[:6]), flush=True)
                print("flagged: build %d step %d, %d replays since the last render, loss %g (non-finite parts: %s; %d non-finite parameters), %d tensors: %s" % (
                    b, k, since, loss.item(), nanparts, wbad, len(bad), bad[:6]), flush=True)
                if wbad:
                    print("build %d: parameters are non-finite, build abandoned" % b, flush=True)
                    break
        if k % EVERY == EVERY - 1:
            n0 = len(conv._NAN_TRACE)
            if KIND == "alloc":            # no model call at all: eager allocations and a few ATen kernels only
                junk = [torch.randn(1 << n, device=dev) for n in (8, 12, 16, 20, 24)]
                junk = [j * 2 for j in junk]
                del junk
            elif KIND == "toggle":         # only the train / eval switch
                trainer.eval(); trainer.train()
            elif KIND == "encoders":       # the conditioning encoders only (no ray march)
                trainer.eval()
                with torch.no_grad():
                    trainer.model_coarse.set_conditional_embedding(latents=trainer.latent_codes[0:1], cond_c=inp["inv_head_T"][:1].reshape(1, -1),
                                                                   **{kk: inp[kk][:1] for kk in ("front_render_cond", "left_render_cond", "right_render_cond")})
                trainer.train()
            else:
                trainer.eval()
                with torch.no_grad():
                    trainer(mode="validation", fidx=None, render_full_img=False, ray_batch=inp["ray_batch"][:1, :256].contiguous(),
                            background_prior=inp["background_prior"][:1, :256].contiguous(), inv_head_T=inp["inv_head_T"][:1],
                            **{kk: inp[kk][:1] for kk in ("front_render_cond", "left_render_cond", "right_render_cond")})
                trainer.train()
            if PROBE and k <= 8:
                torch.cuda.synchronize()
                now = pool_blocks()
                pool_diff("after the eager render behind step %d" % k, snap, now)
                snap = now
            del conv._NAN_TRACE[n0:]                      # (the eager render's own trace entries are not part of the step)
            since = 0
    del run, opt, trainer
print("graph_anomaly_hunt: %s steps, tracer mode %d sentinel %s: %d flagged replays of %d (%d builds x %d steps, render every %d), traced tensors per step %d, %.0f s" % (
    "graphed" if GRAPH else "eager", conv._NAN_MODE, os.environ.get("SENTINEL", "0"), len(events), replays, BUILDS, STEPS, EVERY, len(conv._NAN_TRACE), time.time() - t0))
