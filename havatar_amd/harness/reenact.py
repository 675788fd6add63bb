"""H1 -- reenactment harness (reference: avatarHD_reenactment.py:103-172).

    python avatarHD_reenactment.py --config <yml> --ckpt <pth> --savedir <dir> --split <split.json>

Same flags, same checkpoint keys (`nerf_render`, `latent_codes`, `g_ema`), same per-frame sequence:
test-mode frame (B=1) -> `Trainer(**inp)` (validation mode, full image) -> `SWGAN_unet(styles=[style], condition_img=render[:, 3:])`
-> clip(x*255, 0, 255) uint8 -> `<savedir>/rgb/<fidx>_<view:02d>.png`.
What is different by design: the frame is one hipGraph launch per stage when shapes are static (HAVATAR_GRAPH=0 disables), and
with more than one process (torchrun) the frames of the split are dealt round-robin to the ranks (no data-path collective).
Throughput mode: HAVATAR_FRAME_BATCH=B (default 1 = the reference's sequence) renders B of this rank's frames per call -- the encoders
and the upsampler see a batch, the march takes B x R rays in one launch; files and pixels are those of the one-frame loop (frames of a batch
differ from the same frames rendered alone by the encoders' rounding, 1e-4 on the render; a ragged last batch is rendered at its own size).
"""
import argparse
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import yaml

from . import per_rank_miopen_cache
from ..dataloader import imgio
from ..dataloader.dataloaderSR import Loader
from ..frames import shard_frames
from ..graph import GraphedForward
from ..model.nerf_trainer import Trainer
from ..model.styleUnet import SWGAN_unet
from ..utils.cfgnode import CfgNode
from ..utils.training_util import load_partial_state_dict


class styleUnet_args:
    """The generator hyper-parameters the reference hard-codes in the script (avatarHD_reenactment.py:17-45); only the three the
    inference path reads are kept."""
    latent = 64
    n_mlp = 4
    channel_multiplier = 2


su_args = styleUnet_args()


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config", type=str, default="config/singleview_512_HD_base.yml", help="Path to (.yml) config file.")
    p.add_argument("--ckpt", type=str, default="", help="Path to load saved checkpoint from.")
    p.add_argument("--savedir", type=str, default="./renders/", help="Save images to this directory, if specified.")
    p.add_argument("--split", type=str, default=None, help="Split file (json) listing the frames to render.")
    return p.parse_args(argv)


def build_models(cfg, checkpoint, device):
    render_size, gen_size = cfg.models.StyleUnet.inp_size, cfg.models.StyleUnet.out_size
    nerf_render = Trainer(cfg, 0).requires_grad_(False).to(device)
    img_trans = SWGAN_unet(inp_size=render_size, inp_ch=cfg.models.StyleUnet.inp_ch, out_size=gen_size, out_ch=3,
                           style_dim=su_args.latent, c_dim=0, n_mlp=su_args.n_mlp,
                           channel_multiplier=su_args.channel_multiplier).to(device)
    load_partial_state_dict(nerf_render, checkpoint["nerf_render"], except_keys=["latent_codes"])
    nerf_render.latent_codes = checkpoint["latent_codes"].to(device)
    img_trans.load_state_dict(checkpoint["g_ema"])
    nerf_render.headpose_skin_net.fix_canonical_W()
    return nerf_render.eval(), img_trans.eval()


def frame_inputs(idx, batch, device, size=None, cache=None):
    """The dict the reference builds per frame (:151-159).  With a `camera` record in the batch (device-ray mode) the ray table is
    generated on the device (hav_gen_rays) and the white background prior is a cached constant."""
    if "camera" in batch:
        from ..render import gen_rays
        n = batch["camera"].shape[0]
        cache = cache if cache is not None else {}
        if ("rays", n) not in cache:
            cache[("rays", n)] = torch.empty(n, size * size, 8, device=device)
            cache[("bg", n)] = torch.ones(n, size * size, 3, device=device)
        for b in range(n):
            cam = batch["camera"][b]
            gen_rays(size, size, cam[0:4], cam[4:16], float(cam[16]), float(cam[17]), device, out=cache[("rays", n)][b:b + 1])
        ray_batch, bg = cache[("rays", n)], cache[("bg", n)]
    else:
        rays = batch["mv_rays"]
        ray_batch, bg = rays[..., :-3].to(device), rays[..., -3:].to(device)
    return {"mode": "validation", "fidx": idx, "render_full_img": True,
            "ray_batch": ray_batch, "background_prior": bg,
            # channels-last host images go over PCIe as they are (contiguous, pinned by the loader); the NHWC -> NCHW permutation
            # happens on the device (a strided host-side copy of 3 x 1.8 MB costs more than the frame's GPU work at 128^2)
            "front_render_cond": batch["front_render_cond"].to(device, non_blocking=True).permute(0, 3, 1, 2),
            "left_render_cond": batch["left_render_cond"].to(device, non_blocking=True).permute(0, 3, 1, 2),
            "right_render_cond": batch["right_render_cond"].to(device, non_blocking=True).permute(0, 3, 1, 2),
            "inv_head_T": batch["inv_head_T"].to(device)}


def to_png_array(gen_img):
    """[1,3,H,W] float -> uint8 [H,W,3] RGB exactly as :164 (x*255 in float32, clip, truncating cast) -- done on the device, so
    0.75 MB instead of 3 MB cross PCIe per 512^2 frame."""
    return (gen_img[0].permute(1, 2, 0) * 255).clamp_(0, 255).to(torch.uint8).cpu().numpy()


def main(argv=None, device=None, style=None):
    args = parse_args(argv)
    os.makedirs(os.path.join(args.savedir, "rgb"), exist_ok=True)
    with open(args.config, "r") as f:
        cfg = CfgNode(yaml.load(f, Loader=yaml.FullLoader))
    seed = cfg.experiment.randomseed
    np.random.seed(seed)
    torch.manual_seed(seed)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    per_rank_miopen_cache()
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))       # the reference is GPU-only as well (:133)
    device = torch.device(device)
    if device.type == "cuda":
        device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        torch.cuda.set_device(device)
    print(args.ckpt)
    checkpoint = torch.load(args.ckpt, map_location="cpu")
    nerf_render, img_trans = build_models(cfg, checkpoint, device)
    if style is None:
        style = torch.mean(torch.randn(1000, 1, su_args.latent), dim=0)            # :147, after the constructors consumed the RNG
    style = style.to(device)
    val_loader = Loader(split_file=args.split, mode="test", batch_size=1, options=cfg, down_sample=cfg.dataset.down_sample)
    mine = list(shard_frames(len(val_loader.dataset), rank, world))
    # HIP: the loader ships 18 camera floats per frame and the ray table is built on the device
    val_loader.dataset.device_rays = device.type == "cuda" and os.environ.get("HAVATAR_DEVICE_RAYS", "1") != "0"
    ray_cache = {}
    # this rank's frames only, read ahead by worker processes (the reference reads every frame in the main process)
    fbatch = max(1, int(os.environ.get("HAVATAR_FRAME_BATCH", "1")))
    frames = torch.utils.data.DataLoader(torch.utils.data.Subset(val_loader.dataset, mine), batch_size=fbatch, shuffle=False,
                                         num_workers=int(os.environ.get("HAVATAR_WORKERS", 4)), pin_memory=device.type == "cuda")
    use_graph = device.type == "cuda" and os.environ.get("HAVATAR_GRAPH", "1") != "0"
    graphed, graphed2, written = {}, {}, []            # hipGraphs per batch size (the last batch of a split may be ragged)
    writers = ThreadPoolExecutor(max_workers=int(os.environ.get("HAVATAR_PNG_THREADS", 12)))    # PNG deflate off the critical path (a 1024^2 frame is ~70 ms of zlib on one core)
    pending, ring, in_flight = [], [None, None], None
    t_first = t_loop = None
    with torch.no_grad():
        for idx, val_batch in frames:
            if t_first is None:
                t_first = time.perf_counter()
            elif t_loop is None:
                t_loop = time.perf_counter()             # steady state starts after the first frame (solver search, graph capture)
            n = int(val_batch["fidx"].shape[0])
            inp = frame_inputs(idx, val_batch, device, size=val_loader.dataset.img_h, cache=ray_cache)
            styles = [style if n == 1 else style.expand(n, -1)]
            if use_graph:
                tens = {k_: v for k_, v in inp.items() if torch.is_tensor(v) and k_ != "fidx"}
                if n not in graphed:
                    fixed = {k_: v for k_, v in inp.items() if k_ not in tens}
                    graphed[n] = GraphedForward(lambda fixed=fixed, **kw: nerf_render(**kw, **fixed), tens)
                render, _, _ = graphed[n](**tens)
                cond = {"condition_img": render[:, 3:].contiguous()}
                if n not in graphed2:
                    graphed2[n] = GraphedForward(lambda condition_img, styles=styles: img_trans(styles=styles, condition_img=condition_img), cond)
                gen_imgs = graphed2[n](**cond)
            else:
                render, _, _ = nerf_render(**inp)
                gen_imgs = img_trans(styles=styles, condition_img=render[:, 3:])
            for b in range(n):
                gen_img = gen_imgs[b:b + 1]
                name, k = str(int(val_batch["fidx"][b])), int(val_batch["vidx"][0][b])          # ("vidx" is a one-element list per item: collated to [tensor[n]])
                path = os.path.join(args.savedir, "rgb", f"{name}_{k:02d}.png")
                written.append(path)
                if device.type != "cuda":
                    pending.append(writers.submit(imgio.imwrite_rgb, path, to_png_array(gen_img)))
                    continue
                # two-slot read-back ring: frame n's uint8 image is copied to pinned memory asynchronously and handed to the PNG
                # threads while frame n+1 is being prepared and launched, so the GPU does not idle during host work
                slot = len(written) & 1
                if ring[slot] is None:
                    H2, W2 = gen_img.shape[-2:]
                    ring[slot] = (torch.empty(H2, W2, 3, dtype=torch.uint8, device=device), torch.empty(H2, W2, 3, dtype=torch.uint8).pin_memory(),
                                  torch.cuda.Event())
                dev_u8, host_u8, ev = ring[slot]
                if in_flight is not None and in_flight[1] == slot:                         # (batched frames: the slot's previous image must have left)
                    ring[slot][2].synchronize()
                dev_u8.copy_((gen_img[0].permute(1, 2, 0) * 255).clamp_(0, 255))          # float -> uint8 truncation, as to_png_array
                host_u8.copy_(dev_u8, non_blocking=True)
                ev.record()
                if in_flight is not None:                                                  # finish the PREVIOUS frame now
                    p_path, p_slot = in_flight
                    ring[p_slot][2].synchronize()
                    pending.append(writers.submit(imgio.imwrite_rgb, p_path, ring[p_slot][1].numpy().copy()))
                in_flight = (path, slot)
    if in_flight is not None:
        p_path, p_slot = in_flight
        ring[p_slot][2].synchronize()
        pending.append(writers.submit(imgio.imwrite_rgb, p_path, ring[p_slot][1].numpy().copy()))
    for f in pending:
        f.result()
    writers.shutdown()
    if t_loop is not None and len(written) > 1:
        dt = time.perf_counter() - t_loop
        print("[rank %d] %d frames; first frame %.2f s (solver search + graph capture), then %.1f ms per frame end to end "
              "(read, render %dx%d, upsample %dx%d, PNG)" % (rank, len(written), t_loop - t_first, 1e3 * dt / (len(written) - 1),
                                                            nerf_render.render_size, nerf_render.render_size, gen_img.shape[-1], gen_img.shape[-2]))
    print("Done!")
    return written


if __name__ == "__main__":
    main()
