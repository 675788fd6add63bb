#!/usr/bin/env python3
"""Localise capture problems of the training step: python tools/debug_train_graph.py <stage>  (fwd | bwd | opt)."""
import faulthandler, os, sys, tempfile
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from havatar_amd import synth
from havatar_amd.dataloader.dataloader import Loader
from havatar_amd.harness import train
from havatar_amd.model.nerf_trainer import Trainer
from havatar_amd.utils.cfgnode import CfgNode

stage = sys.argv[1]
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
opt = train.make_optimizer(cfg, trainer, True)
runner = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=True)
inp, target, mask = train.step_inputs(idx, batch, dev)
tens = {k: v.to(dev) for k, v in inp.items() if torch.is_tensor(v)}
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, aux = runner._loss(target, mask, **tens)
        loss.backward()
        opt.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warmup ok", loss.item(), flush=True)
opt.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    loss, aux = runner._loss(target, mask, **tens)
    print("captured forward", flush=True)
    if stage in ("bwd", "opt"):
        loss.backward()
        print("captured backward", flush=True)
    if stage == "opt":
        opt.step()
        print("captured optimizer", flush=True)
print("capture closed", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed", stage, loss.item(), flush=True)
