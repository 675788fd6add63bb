"""H2 -- stage-one training harness (reference: train_avatar.py:30-319).

    python train_avatar.py --logdir <dir> --datadir <dir with sv_v31_all.json> --config <yml> [--ckpt <ckpt>]

Same flags, loss formula, optimiser / learning-rate schedule and checkpoint keys (`iter`, `optimizer_state_dict`, `loss`, `psnr`,
`trainer_state_dict`).  The step runs `Trainer(**inp)` in train mode; with autograd on, the Trainer takes the PyTorch statement
of the path (the fused HIP kernel is forward-only), while the encoders' custom ops are the HIP kernels with their
first/second-order autograd.  Validation frames are rendered through the fused kernel (no-grad).
LPIPS (patch_rgb) needs the `lpips` package and its VGG weights; when they are absent the harness refuses to start unless
`--percep none` is given, which drops that one term and says so.
"""
import argparse
import datetime
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from ..dataloader import imgio
from ..dataloader.dataloader import Loader
from ..model.nerf_trainer import Trainer
from ..utils.cfgnode import CfgNode
from ..utils.training_util import mse2psnr


def lpips_loss(img0, img1, lpips_fn):
    """[B,H,W,3] in [0,1] -> mean LPIPS (train_avatar.py:24-29)."""
    img0 = img0.permute(0, 3, 1, 2) * 2.0 - 1.0
    img1 = img1.permute(0, 3, 1, 2) * 2.0 - 1.0
    return lpips_fn.forward(img0, img1).mean()


def skin_weight_smoothness(trainer):
    """Mean absolute 6-neighbour difference of the second blend-weight channel (train_avatar.py:123-129)."""
    net = trainer.headpose_skin_net
    # (the same evaluation the march samples: one decoder pass per step; a reference Trainer has no volume_once and re-evaluates)
    vol = (net.volume_once() if hasattr(net, "volume_once") else net.canonical_Wvolume())[0, 1]
    core = vol[1:-1, 1:-1, 1:-1]
    nbrs = [vol[:-2, 1:-1, 1:-1], vol[2:, 1:-1, 1:-1], vol[1:-1, 2:, 1:-1], vol[1:-1, :-2, 1:-1], vol[1:-1, 1:-1, 2:], vol[1:-1, 1:-1, :-2]]
    return torch.mean(sum(torch.abs(core - v) for v in nbrs) / 6.0)


def step_inputs(idx, batch, device):
    """train_avatar.py:108-121: mv_rays [B,R,12] = rays(8) | background(3) | mask(1)."""
    mv_rays = batch["mv_rays"].to(device)
    inp = {"mode": "train", "fidx": idx, "render_full_img": False,
           "ray_batch": mv_rays[..., :-4], "background_prior": mv_rays[..., -4:-1],
           "front_render_cond": batch["front_render_cond"].permute(0, 3, 1, 2).to(device),
           "left_render_cond": batch["left_render_cond"].permute(0, 3, 1, 2).to(device),
           "right_render_cond": batch["right_render_cond"].permute(0, 3, 1, 2).to(device),
           "inv_head_T": batch["inv_head_T"].to(device)}
    return inp, batch["mv_rays_gt_color"].to(device), mv_rays[..., -1:]


def training_loss(trainer, cfg, inp, target, ray_mask, rgb_loss_func, percep_loss_fn=None):
    """One forward of the step and its scalar loss (train_avatar.py:121-146).  Returns (loss, parts, psnr)."""
    rgb_coarse, _, acc_coarse, weights, rgb_fine, _, acc_fine, latent_code_loss = trainer(**inp)
    sw_grad_loss = skin_weight_smoothness(trainer)
    if rgb_coarse.is_cuda:
        from ..native import conv as _nconv
        _nconv._trace("Trainer outputs rgb_c,acc_c,weights,rgb_f,acc_f", rgb_coarse, acc_coarse, weights, rgb_fine, acc_fine)      # (development aid; a no-op unless HAVATAR_NAN_TRACE)
    parts = {"coarse_loss": rgb_loss_func(rgb_coarse[..., :3], target[..., :3]),
             "mask_coarse_loss": F.binary_cross_entropy(acc_coarse.clip(1e-3, 1.0 - 1e-3), ray_mask)}
    if rgb_fine is not None:
        parts["fine_loss"] = rgb_loss_func(rgb_fine[..., :3], target[..., :3])
        parts["mask_fine_loss"] = F.binary_cross_entropy(acc_fine.clip(1e-3, 1.0 - 1e-3), ray_mask)
    if percep_loss_fn is not None:
        patch = rgb_coarse[..., :3] if rgb_fine is None else rgb_fine[..., :3]
        s = int(patch.shape[1] ** 0.5)
        parts["patch_percep_loss"] = lpips_loss(patch.reshape(patch.shape[0], s, s, 3), target[..., :3].reshape(patch.shape[0], s, s, 3),
                                                percep_loss_fn)
    mw = cfg.experiment.mask_weight
    loss = parts["coarse_loss"] + mw * parts["mask_coarse_loss"] + (parts["patch_percep_loss"] * 0.05 if percep_loss_fn is not None else 0.0)
    if rgb_fine is not None:
        loss = loss + (parts["fine_loss"] + mw * parts["mask_fine_loss"])
    loss = loss + latent_code_loss + sw_grad_loss * 1e-4
    parts.update(code_loss=latent_code_loss, sw_grad_loss=sw_grad_loss)
    best = rgb_coarse if rgb_fine is None else rgb_fine
    mse = F.mse_loss(best[..., :3], target[..., :3]).detach()
    if mse.is_cuda and torch.cuda.is_current_stream_capturing():
        return loss, parts, mse                       # a captured step cannot read it back: the caller converts after the replay
    return loss, parts, mse2psnr(mse.item())


def graph_training_enabled(device, percep_loss_fn=None):
    """The step runs as one hipGraph launch (graph.GraphedTrainStep) on HIP devices unless HAVATAR_TRAIN_GRAPH=0."""
    return (torch.device(device).type == "cuda" and percep_loss_fn is None and os.environ.get("HAVATAR_TRAIN_GRAPH", "1") != "0"
            and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0")       # the ATen statement of the march reads host tensors


def make_optimizer(cfg, trainer, graph):
    kw = {}
    if graph and cfg.optimizer.type in ("Adam", "AdamW", "NAdam", "RAdam", "Adamax", "Adagrad", "RMSprop", "SGD", "ASGD", "Adadelta", "Rprop"):
        # capturable: the step counter and the learning rate live on the device, so the optimiser update can be replayed
        kw = {"capturable": True} if cfg.optimizer.type != "SGD" else {}
        if cfg.optimizer.type in ("Adam", "AdamW") and os.environ.get("HAVATAR_FUSED_ADAM", "1") != "0":
            kw["fused"] = True            # one multi-tensor kernel per parameter bucket instead of ~8 foreach passes (same update formula)
    return getattr(torch.optim, cfg.optimizer.type)([{"params": list(trainer.parameters())}], lr=cfg.optimizer.lr, **kw)


def portable_optimizer_state(optimizer):
    """optimizer.state_dict() with plain-float learning rates: what the reference's checkpoints hold (train_avatar.py:303-316), and
    what every loader -- eager, graph mode, the reference itself -- restores correctly."""
    sd = optimizer.state_dict()
    for g in sd["param_groups"]:
        if torch.is_tensor(g.get("lr")):
            g["lr"] = float(g["lr"])
    return sd


def set_learning_rate(optimizer, lr_new):
    for g in optimizer.param_groups:
        if torch.is_tensor(g["lr"]):
            g["lr"].fill_(lr_new)                     # device-resident rate of a capturable optimiser: the graph reads it
        else:
            g["lr"] = lr_new


def enable_determinism(on=True):
    """HAVATAR_DETERMINISTIC=1: a bit-reproducible optimisation step (the same weights, batch and seeds give the same bits in every gradient,
    run after run; tools/step_determinism.py: 0 of 154 tensors differ, against 141 without).  Three switches: this library's field-input
    scatter sums in 64-bit fixed point (native/train_ops.py::deterministic: the float atomics of the default route land in a different
    order every time), MIOpen is restricted to deterministic solvers (torch.backends.cudnn.deterministic: the default solver of the 16^2
    convolutions is what makes even the FORWARD differ between runs), ATen's index / scatter adds take their deterministic forms."""
    if os.environ.get("HAVATAR_DET_MIOPEN", "1") != "0":          # (A/B: the parts one by one)
        torch.backends.cudnn.deterministic = bool(on)
    if os.environ.get("HAVATAR_DET_ATEN", "1") != "0":
        torch.use_deterministic_algorithms(bool(on), warn_only=True)


def deterministic_requested():
    return os.environ.get("HAVATAR_DETERMINISTIC", "0") == "1"


class StepRunner:
    """forward + backward + optimiser update of one batch; eager for the first `eager_steps` calls (solver selection, lazy
    optimiser state) and for odd batch shapes, one hipGraph replay otherwise."""

    def __init__(self, trainer, cfg, optimizer, rgb_loss_func, percep_loss_fn=None, graph=False, eager_steps=2):
        self.trainer, self.cfg, self.optimizer, self.rgb_loss_func, self.percep = trainer, cfg, optimizer, rgb_loss_func, percep_loss_fn
        self.graph, self.eager_left, self.graphed, self.side = graph, eager_steps, None, None
        if deterministic_requested():
            enable_determinism()
        if graph:
            dev = next(trainer.parameters()).device
            for g in optimizer.param_groups:          # capturable Adam reads a DEVICE tensor learning rate inside the graph
                # always re-made: a rate restored from a checkpoint is a float or a CPU tensor (map_location="cpu"), and a CPU
                # 0-dim tensor would be baked into the captured graph as a constant -- set_learning_rate() would never reach it
                if not (torch.is_tensor(g["lr"]) and g["lr"].device == dev):
                    g["lr"] = torch.as_tensor(float(g["lr"]), dtype=torch.float32, device=dev)
                if "capturable" in g:
                    g["capturable"] = True            # a checkpoint written by an eager run (or by the reference) restores False
            for st in optimizer.state.values():
                if torch.is_tensor(st.get("step")) and st["step"].device != dev:
                    st["step"] = st["step"].to(dev)

    def _loss(self, target, ray_mask, **inp):
        t = self.trainer
        full = dict(inp, mode="train", render_full_img=False)
        loss, parts, mse = training_loss(t, self.cfg, full, target, ray_mask, self.rgb_loss_func, self.percep)
        return loss, dict(parts, _mse=mse)

    def __call__(self, inp, target, ray_mask):
        """-> (loss, parts, psnr); gradients are consumed (optimizer.step + zero_grad) inside."""
        dev = target.device
        tens = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items() if torch.is_tensor(v)}
        if self.graph and self.eager_left <= 0 and (self.graphed is None or self.graphed.matches(dict(tens, target=target, ray_mask=ray_mask))):
            from ..graph import GraphedTrainStep
            if self.graphed is None:
                from ..native import conv as _nconv
                if _nconv._NAN_TRACE is not None:
                    _nconv._NAN_TRACE.clear()          # (development aid: keep the flags of the captured step only)
                    _nconv.nan_trace_reset(dev)        # (its flag buffer exists before the capture: never a graph-pool block)
                if self.side is None:
                    self.side = torch.cuda.Stream(device=dev)
                self.side.wait_stream(torch.cuda.current_stream(dev))
                self.graphed = GraphedTrainStep(lambda target, ray_mask, **kw: self._loss(target, ray_mask, **kw), self.optimizer,
                                                dict(tens, target=target, ray_mask=ray_mask),
                                                stream=None if os.environ.get("HAVATAR_GRAPH_CAPTURE_STREAM") == "own" else self.side)          # ("own": the old, forking capture; A/B only)
                torch.cuda.current_stream(dev).wait_stream(self.side)
            loss, aux = self.graphed(**tens, target=target, ray_mask=ray_mask)
            parts = {k: v for k, v in aux.items() if k != "_mse"}
            return loss, parts, mse2psnr(aux["_mse"].item())
        self.eager_left -= 1
        if not self.graph:
            loss, parts, psnr = training_loss(self.trainer, self.cfg, inp, target, ray_mask, self.rgb_loss_func, self.percep)
            loss.backward()
            self.optimizer.step()
            self.optimizer.zero_grad()
            return loss, parts, psnr
        # the eager steps that precede a capture run on a side stream, as the capture itself will: gradient buffers and optimiser
        # state created on the default stream make the captured backward crash (AccumulateGrad stream mismatch)
        from ..graph import bump_weights_epoch
        cur = torch.cuda.current_stream(dev)
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.optimizer.zero_grad(set_to_none=True)
            loss, parts, psnr = training_loss(self.trainer, self.cfg, inp, target, ray_mask, self.rgb_loss_func, self.percep)
            loss.backward()
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
        cur.wait_stream(self.side)
        bump_weights_epoch()
        return loss, parts, psnr


def learning_rate(cfg, i):
    """train_avatar.py:153-154: exponential decay per 1000*lr_decay steps, floored at 5e-5."""
    return max(cfg.optimizer.lr * (cfg.scheduler.lr_decay_factor ** (i / (cfg.scheduler.lr_decay * 1000))), 5e-5)


def validate(trainer, cfg, val_batch, img_h, img_w, device, rgb_loss_func):
    """Full-frame validation render in chunks of nerf.validation.chunksize rays (train_avatar.py:183-216)."""
    _, batch = val_batch
    rays = batch["mv_rays"][0].reshape(-1, batch["mv_rays"][0].shape[-1])
    inp = {"mode": "validation", "fidx": None, "render_full_img": False,
           "front_render_cond": batch["front_render_cond"].permute(0, 3, 1, 2).to(device),
           "left_render_cond": batch["left_render_cond"].permute(0, 3, 1, 2).to(device),
           "right_render_cond": batch["right_render_cond"].permute(0, 3, 1, 2).to(device),
           "inv_head_T": batch["inv_head_T"].to(device)}
    n, group = rays.shape[0], cfg.nerf.validation.chunksize
    coarse, fine = [], []
    for s in range(0, n, group):
        inp.update(ray_batch=rays[s:s + group][..., :-3].to(device).unsqueeze(0),
                   background_prior=rays[s:s + group][..., -3:].to(device).unsqueeze(0))
        rgb_c, _, _, _, rgb_f, _, _, _ = trainer(**inp)
        coarse.append(rgb_c[0][..., :3].detach().cpu())
        if rgb_f is not None:
            fine.append(rgb_f[0][..., :3].detach().cpu())
    views = n // (img_h * img_w)
    target = batch["mv_rays_gt_color"][0].reshape(views, img_h, img_w, 3)
    img = torch.cat(fine if fine else coarse, 0).reshape(views, img_h, img_w, 3)
    loss = rgb_loss_func(img, target)
    return img, target, loss.item(), mse2psnr(F.mse_loss(img, target).item())


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    add_image = add_scalar


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--logdir", type=str, required=True)
    p.add_argument("--datadir", type=str, required=True)
    p.add_argument("--config", type=str, default="config/singleview_512_base.yml", help="Path to (.yml) config file.")
    p.add_argument("--ckpt", type=str, default="", help="Path to load saved checkpoint from.")
    p.add_argument("--percep", type=str, default="lpips", choices=["lpips", "none"], help="'none' drops the LPIPS patch term (no lpips package / weights)")
    p.add_argument("--max-steps", type=int, default=0, help="stop after this many optimisation steps (0 = experiment.train_iters)")
    return p.parse_args(argv)


def main(argv=None, device=None):
    args = parse_args(argv)
    now = datetime.datetime.now()
    with open(args.config, "r") as f:
        cfg = CfgNode(yaml.load(f, Loader=yaml.FullLoader))
    seed = cfg.experiment.randomseed
    np.random.seed(seed)
    torch.manual_seed(seed)
    device = torch.device(device if device is not None else "cuda")
    percep_loss_fn = None
    if cfg.experiment.patch_rgb and args.percep == "lpips":
        try:
            import lpips
        except ImportError as e:
            raise RuntimeError("experiment.patch_rgb needs the `lpips` package (VGG weights); install it or pass --percep none") from e
        percep_loss_fn = lpips.LPIPS(net="vgg").to(device)
    elif cfg.experiment.patch_rgb:
        print("[train] --percep none: the 0.05 * LPIPS(patch) term of the reference loss is dropped")
    rgb_loss_func = F.mse_loss if cfg.experiment.rgb_loss == "mse" else F.l1_loss
    split = os.path.join(args.datadir, "sv_v31_all.json")
    workers = int(os.environ.get("HAVATAR_WORKERS", 8))
    train_loader = Loader(split_file=split, mode="train", batch_size=2, num_workers=workers, down_sample=cfg.dataset.down_sample,
                          options=cfg, white_bg=True)
    val_loader = Loader(split_file=split, shuffle=True, mode="val", batch_size=1, num_workers=min(workers, 1), down_sample=1.0,
                        options=cfg, white_bg=True)
    val_data = enumerate(val_loader)
    trainer = Trainer(cfg, len(train_loader.dataset)).to(device)
    use_graph = graph_training_enabled(device, percep_loss_fn)
    optimizer = make_optimizer(cfg, trainer, use_graph)
    os.makedirs(args.logdir, exist_ok=True)
    try:
        from torch.utils.tensorboard import SummaryWriter
        writer = SummaryWriter(args.logdir)
    except Exception:  # tensorboard is optional here
        writer = _NullWriter()
    with open(os.path.join(args.logdir, "config_%s.tar.yml" % now.strftime("%Y_%m_%d_%H_%M_%S")), "w") as f:
        f.write(cfg.dump())
    start_iter = -1
    if len(args.ckpt) > 0:
        assert os.path.exists(args.ckpt)
        checkpoint = torch.load(args.ckpt, map_location="cpu")
        trainer.load_state_dict(checkpoint["trainer_state_dict"])
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        start_iter = checkpoint["iter"]
    elif trainer.headpose_skin_net is not None:
        trainer.headpose_skin_net.pretrain_wc(num_iter=int(os.environ.get("HAVATAR_PRETRAIN_WC", 3000)), vol_thr=cfg.models.coarse.Head_bounding)
    i, steps_done = start_iter, 0
    loss, psnr = None, None
    run_step = StepRunner(trainer, cfg, optimizer, rgb_loss_func, percep_loss_fn, graph=use_graph)
    while i < cfg.experiment.train_iters:
        trainer.train()
        t0 = time.time()
        for idx, batch in train_loader:
            i += 1
            inp, target, ray_mask = step_inputs(idx, batch, device)
            loss, parts, psnr = run_step(inp, target, ray_mask)
            if os.environ.get("HAVATAR_NAN_TRACE"):
                from ..native import conv as _nconv
                bad = [(k, n) for k, (n, f) in enumerate(_nconv._NAN_TRACE or []) if not bool(f)]
                if bad:
                    print("[nan-trace] iter %d (loss %s): %d traced tensors, %d non-finite; first (index, name): %s" % (
                        i, float(loss), len(_nconv._NAN_TRACE or []), len(bad), bad[:8]))
            lr_new = learning_rate(cfg, i)
            set_learning_rate(optimizer, lr_new)
            if i % cfg.experiment.print_every == 0 or i == cfg.experiment.train_iters - 1:
                print("[TRAIN] Iter: %d Loss: %.06f PSNR: %.06f LatentReg: %.04f e-5 LR: %.02f e-5 TIME: %.02f" % (
                    i, loss.item(), psnr, 1e5 * parts["code_loss"].item(), 1e5 * lr_new, (time.time() - t0) / cfg.experiment.print_every))
                t0 = time.time()
            for k, v in parts.items():
                writer.add_scalar("train/" + k, v.item(), i)
            writer.add_scalar("train/psnr", psnr, i)
            if i == start_iter + 1 or i % cfg.experiment.validate_every == 0:
                trainer.eval()
                with torch.no_grad():
                    try:
                        vb = next(val_data)
                    except StopIteration:
                        val_data = enumerate(val_loader)
                        vb = next(val_data)
                    img, target_img, vloss, vpsnr = validate(trainer, cfg, vb[1], val_loader.dataset.img_h, val_loader.dataset.img_w, device, rgb_loss_func)
                writer.add_scalar("validation/psnr", vpsnr, i)
                imgio.imwrite_rgb(os.path.join(args.logdir, "val_%05d.png" % i),
                                  (torch.cat([img[0], target_img[0]], 1).clamp(0, 1) * 255).round().byte().numpy())
                print("Validation loss: %06f Validation PSNR: %06f" % (vloss, vpsnr))
                trainer.train()
            if i % cfg.experiment.save_every == 0 or i == cfg.experiment.train_iters - 1 or i == start_iter + 1:
                torch.save({"iter": i, "optimizer_state_dict": portable_optimizer_state(optimizer), "loss": loss.detach().clone(), "psnr": psnr,
                            "trainer_state_dict": trainer.state_dict()}, os.path.join(args.logdir, "checkpoint" + str(i).zfill(5) + ".ckpt"))
                trainer.headpose_skin_net.visualize_motion_weight_vol(os.path.join(args.logdir, "vis_motionWeightVol" + str(i).zfill(5) + ".obj"))
                print("================== Saved Checkpoint =================")
            steps_done += 1
            if (args.max_steps and steps_done >= args.max_steps) or i >= cfg.experiment.train_iters:
                print("Done!")
                return i
    print("Done!")
    return i


if __name__ == "__main__":
    np.random.seed(999)
    torch.random.manual_seed(999)
    main()
