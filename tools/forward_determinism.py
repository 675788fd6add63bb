#!/usr/bin/env python3
"""Which module of the training forward is not bit-reproducible?  The same forward (same weights, batch, seeds) twice; a forward hook on every
leaf-ish module records a checksum of its output (int64 sum of the fp32 bit patterns); the first modules whose checksums differ are printed in
execution order."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from havatar_amd import synth
from havatar_amd.dataloader.dataloader import Loader
from havatar_amd.harness import train
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

dev = torch.device("cuda:0")
if os.environ.get("CUDNN_DET") == "1":          # MIOpen: deterministic solvers only
    torch.backends.cudnn.deterministic = True
if os.environ.get("TORCH_DET") == "1":
    torch.use_deterministic_algorithms(True, warn_only=True)
tmp = tempfile.mkdtemp()
split = synth.write_dataset(tmp, n_frames=2, img_res=128)
cfg = CfgNode(synth.harness_config(perturb=True, noise_std=0.1))
np.random.seed(3)
tl = Loader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg, white_bg=True, shuffle=False)
idx, batch = next(iter(tl))
torch.manual_seed(11)
trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(dev).train()
inp, target, mask = train.step_inputs(idx, batch, dev)
log = []


def cks(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item()) if torch.is_tensor(t) and t.dtype == torch.float32 else None


def hook(name):
    def fn(mod, args, out):
        o = out[0] if isinstance(out, (tuple, list)) else out
        log.append((name, type(mod).__name__, cks(o)))
    return fn


for n, m in trainer.named_modules():
    if n:
        m.register_forward_hook(hook(n))
runs = []
for k in range(4):
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    log.clear()
    with torch.enable_grad():
        loss, parts, _ = train.training_loss(trainer, cfg, inp, target, mask, torch.nn.functional.mse_loss, None)
    torch.cuda.synchronize()
    runs.append(list(log))
    print("run %d: %d module outputs, loss %.9f" % (k, len(log), loss.item()), flush=True)
first = []
runs = runs[1:]          # (run 0 also evaluates what later runs take from caches: different module sequence)
for i, (a, b, c) in enumerate(zip(*runs)):
    assert a[0] == b[0] == c[0]
    if not (a[2] == b[2] == c[2]):
        first.append((i, a[0], a[1]))
print("forward_determinism: %d of %d module outputs differ between the runs; the first ones in execution order:" % (len(first), len(runs[0])))
for i, n, t in first[:12]:
    print("   #%d %s (%s)" % (i, n, t))
