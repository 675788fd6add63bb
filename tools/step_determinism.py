#!/usr/bin/env python3
"""Is the optimisation step bit-reproducible?  The same step (same weights, same batch, same seeds) RUNS times in one process; every
parameter gradient of run k is compared bitwise with run 0.  HAVATAR_DETERMINISTIC=1 = harness/train.py::enable_determinism (fixed-point
scatter of the field inputs + MIOpen's deterministic solvers + ATen's deterministic index adds); SCATTER_ONLY=1 / CUDNN_DET=1 / TORCH_DET=1
switch the parts on one by one.  What still differs (if anything) is listed by parameter name with its largest relative difference."""
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

RUNS = int(os.environ.get("RUNS", "4"))
dev = torch.device("cuda:0")
if os.environ.get("CUDNN_DET") == "1":          # (single switches, to see which one a difference needs)
    torch.backends.cudnn.deterministic = True
if os.environ.get("TORCH_DET") == "1":
    torch.use_deterministic_algorithms(True, warn_only=True)
if train.deterministic_requested() and os.environ.get("SCATTER_ONLY") != "1":
    train.enable_determinism()          # what HAVATAR_DETERMINISTIC=1 means for the training harness
tmp = tempfile.mkdtemp()
split = synth.write_dataset(tmp, n_frames=2, img_res=128)
cfg = CfgNode(synth.harness_config(perturb=True, noise_std=0.1))
np.random.seed(3)
tl = Loader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg, white_bg=True, shuffle=False)
idx, batch = next(iter(tl))
torch.manual_seed(11)
trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(dev).train()
inp, target, mask = train.step_inputs(idx, batch, dev)
ref, worst = None, {}
for k in range(RUNS):
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    for p in trainer.parameters():
        p.grad = None
    loss, parts, _ = train.training_loss(trainer, cfg, inp, target, mask, torch.nn.functional.mse_loss, None)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in trainer.named_parameters() if p.grad is not None}
    grads["<loss>"] = loss.detach().clone().reshape(1)
    grads["<planes grad>"] = None
    print("run %d: loss %.9f  parts %s" % (k, loss.item(), {n_: round(float(v), 9) for n_, v in parts.items() if torch.is_tensor(v)}), flush=True)
    del grads["<planes grad>"]
    if ref is None:
        ref = grads
        continue
    for n, g in grads.items():
        if not torch.equal(g, ref[n]):
            rel = float((g - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-30))
            worst[n] = max(worst.get(n, 0.0), rel)
print("step_determinism: HAVATAR_DETERMINISTIC=%s, %d runs, %d tensors compared: %d differ between runs" % (
    os.environ.get("HAVATAR_DETERMINISTIC", "0"), RUNS, len(ref), len(worst)))
for n, r in sorted(worst.items(), key=lambda kv: -kv[1])[:40]:
    print("   %-70s max relative difference %.2e" % (n, r))
