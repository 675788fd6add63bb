#!/usr/bin/env python3
"""ATen-op level attribution of one eager training step (which Python-visible ops the ~2 600 tiny kernels of cfg5 come from):
torch.profiler, grouped by op name and by calling module.  Run on the GPU box."""
import os
import sys
import tempfile

os.environ["HAVATAR_TRAIN_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from havatar_amd import synth
from havatar_amd.dataloader.dataloader import Loader
from havatar_amd.harness import train
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
split = synth.write_dataset(tmp, n_frames=2, img_res=512)
cfgd = synth.harness_config(render_size=128, gen_size=512, img_res=512, perturb=True, noise_std=0.1, rays=4096)
cfgd["experiment"]["patch_rgb"] = True
cfg = CfgNode(cfgd)
np.random.seed(0); torch.manual_seed(0)
tl = Loader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg, white_bg=True, shuffle=False)
idx, batch = next(iter(tl))
trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(dev).train()
opt = train.make_optimizer(cfg, trainer, False)
run = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=False)
inp, target, mask = train.step_inputs(idx, batch, dev)
for _ in range(3):
    run(inp, target, mask)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_modules=True, with_stack=False, record_shapes=False) as prof:
    run(inp, target, mask)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
# attribute device time of tiny kernels to the innermost nn.Module
try:
    ev = prof.key_averages(group_by_stack_n=0)
except Exception:
    ev = None
