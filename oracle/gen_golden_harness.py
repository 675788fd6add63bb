#!/usr/bin/env python3
"""Golden vectors for the harness rows H1 / H2: run the REFERENCE modules (imported from /root/reference, build
container only) through the per-frame sequence of avatarHD_reenactment.py:147-166 and the per-step sequence of
train_avatar.py:106-148 and store the results in tests/golden/harness.npz.

What is the reference here and what is not: `Trainer`, `SWGAN_unet`, autograd through them and the RNG stream are the
reference's.  The two scripts themselves cannot be imported (top-level `import cv2`, `lpips`, tensorboard, and their body is
`main()`), so the frames are read by this repo's dataset readers (cv2-free; pinned against the REFERENCE readers by
tests/golden/small.npz, oracle/gen_golden_reader.py).  The scripts' per-frame / per-step STATEMENTS are executed from their text
(read from /root/reference at generation time, never copied): avatarHD_reenactment.py:153-170 produces the H1 PNG arrays and file
names (asserted equal to this repo's `to_png_array` path), train_avatar.py:108-158 produces the H2-ref step (loss expression,
backward / step / zero_grad order, learning-rate decay).  The per-part loss values of H2 (`h2_*_part_*`) additionally evaluate
havatar_amd.harness.train.training_loss on the REFERENCE trainer object.
Weights are key-derived (synth.fill_state_dict), noise strengths zeroed so that unpinned per-call noise cannot matter."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from gen_golden import import_reference  # noqa: E402


def main():
    from havatar_amd import synth
    from havatar_amd.dataloader.dataloader import Loader as TrainLoader
    from havatar_amd.dataloader.dataloaderSR import Loader as SRLoader
    from havatar_amd.harness import reenact, train
    torch, Trainer, _ = import_reference()
    from model.styleUnet import SWGAN_unet
    from utils.cfgnode import CfgNode
    from utils.training_util import load_partial_state_dict
    out = {}
    tmp = tempfile.mkdtemp()
    split = synth.write_dataset(tmp, n_frames=2, img_res=128)

    # ---------------- H1: reenactment -------------------------------------------------------------------------------
    cfg = CfgNode(synth.harness_config())
    seed = cfg.experiment.randomseed
    np.random.seed(seed)
    torch.manual_seed(seed)
    nerf_render = Trainer(cfg, 0).requires_grad_(False)
    img_trans = SWGAN_unet(inp_size=32, inp_ch=cfg.models.StyleUnet.inp_ch, out_size=128, out_ch=3, style_dim=64, c_dim=0, n_mlp=4,
                           channel_multiplier=2)
    # the checkpoint a user would hand over: key-derived weights, built on throw-away twins so the RNG stream above is untouched
    g = torch.Generator()
    rng_state = torch.get_rng_state()
    tw = synth.fill_state_dict(Trainer(cfg, 3))
    sw = synth.fill_state_dict(SWGAN_unet(inp_size=32, inp_ch=cfg.models.StyleUnet.inp_ch, out_size=128, out_ch=3, style_dim=64, c_dim=0,
                                          n_mlp=4, channel_multiplier=2), seed=1)
    torch.set_rng_state(rng_state)
    ckpt = {"nerf_render": synth.zero_noise_weights({k: v.clone() for k, v in tw.state_dict().items()}),
            "latent_codes": tw.state_dict()["latent_codes"].clone(),
            "g_ema": synth.zero_noise_weights({k: v.clone() for k, v in sw.state_dict().items()})}
    load_partial_state_dict(nerf_render, ckpt["nerf_render"], except_keys=["latent_codes"])
    nerf_render.latent_codes = ckpt["latent_codes"]
    img_trans.load_state_dict(ckpt["g_ema"])
    nerf_render.headpose_skin_net.fix_canonical_W()
    nerf_render.eval(); img_trans.eval()
    style = torch.mean(torch.randn(1000, 1, 64), dim=0)
    out["h1_style"] = style.numpy()
    loader = SRLoader(split_file=split, mode="test", batch_size=1, options=cfg, down_sample=cfg.dataset.down_sample)
    # H1-ref: the reference SCRIPT's own per-frame statements (avatarHD_reenactment.py:153-170, from `name = ...` to `cv2.imwrite`),
    # read from the checkout at generation time and exec'd; cv2 is a two-function stand-in that captures the image it is given.
    import textwrap
    import types
    rsrc = open(os.path.join("/root/reference", "avatarHD_reenactment.py")).read().split("\n")
    ra = next(i for i, ln in enumerate(rsrc) if "name = str(int(val_batch['fidx'][0].numpy()))" in ln)
    rb = next(i for i, ln in enumerate(rsrc) if "cv2.imwrite(" in ln)
    frame_code = textwrap.dedent("\n".join(rsrc[ra:rb + 1]))
    written = {}
    cv2s = types.SimpleNamespace(COLOR_RGB2BGR=4, cvtColor=lambda img, code: img[:, :, ::-1],
                                 imwrite=lambda path, bgr: written.__setitem__(os.path.basename(path), np.ascontiguousarray(bgr[:, :, ::-1])))
    with torch.no_grad():
        for idx, batch in loader:
            ns = {"val_batch": batch, "idx": idx, "device": "cpu", "nerf_render": nerf_render, "img_trans": img_trans, "style": style,
                  "np": np, "os": os, "cv2": cv2s, "torch": torch, "configargs": types.SimpleNamespace(savedir="/tmp/h1ref")}
            exec(frame_code, ns)
    with torch.no_grad():
        for idx, batch in loader:
            inp = reenact.frame_inputs(idx, batch, "cpu")
            render, mask, _ = nerf_render(**inp)
            gen = img_trans(styles=[style], condition_img=render[:, 3:])
            k = int(batch["fidx"][0])
            out["h1_render_%d" % k] = render.numpy()
            out["h1_mask_%d" % k] = mask.numpy()
            out["h1_gen_%d" % k] = gen.numpy()
            out["h1_png_%d" % k] = reenact.to_png_array(gen)
            fname = "%s_%02d.png" % (str(int(batch["fidx"][0])), int(batch["vidx"][0]))
            assert np.array_equal(written[fname], out["h1_png_%d" % k]), "the reference script's frame statements and this repo's disagree"
            out["h1ref_file_%d" % k] = np.array(fname)
            print("H1 frame", k, "render", tuple(render.shape), "acc", float(mask.min()), float(mask.max()), "gen range",
                  float(gen.min()), float(gen.max()))

    # ---------------- H2: one training step ------------------------------------------------------------------------
    for tag, perturb, noise in (("det", False, 0.0), ("rnd", True, 0.1)):
        cfg = CfgNode(synth.harness_config(perturb=perturb, noise_std=noise))
        np.random.seed(7)
        tl = TrainLoader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                         white_bg=True, shuffle=False)
        idx, batch = next(iter(tl))
        torch.manual_seed(5)          # pins the construction-time random StyleGAN_zxc.zero_noise[0] (not in the state_dict, SURVEY B-3)
        trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset)))
        trainer.train()
        inp, target, ray_mask = train.step_inputs(idx, batch, "cpu")
        torch.manual_seed(123)
        loss, parts, psnr = train.training_loss(trainer, cfg, inp, target, ray_mask, torch.nn.functional.mse_loss)
        trainer.model_coarse.triPlane_embeddings.retain_grad()
        loss.backward()
        if tag == "det":
            out["h2_mv_rays"] = batch["mv_rays"].numpy()
            out["h2_target"] = batch["mv_rays_gt_color"].numpy()
            out["h2_fidx"] = idx.numpy()
        out["h2_%s_loss" % tag] = np.array(loss.item())
        out["h2_%s_psnr" % tag] = np.array(psnr)
        for k, v in parts.items():
            out["h2_%s_part_%s" % (tag, k)] = np.array(v.item())
        params = dict(trainer.named_parameters())
        gp = trainer.model_coarse.triPlane_embeddings.grad
        out["h2_%s_grad_planes_slice" % tag] = gp[:, :, ::8, ::16, ::16].numpy()
        out["h2_%s_grad_planes_cks" % tag] = np.array([gp.double().sum().item(), gp.double().abs().sum().item(), gp.double().abs().max().item()])
        names = ["model_coarse.layers_xyz.0.weight", "model_coarse.layers_xyz.1.weight", "model_coarse.fc_alpha.weight",
                 "model_coarse.fc_rgbFeat.weight", "model_coarse.fc_rgb.weight", "latent_codes"]
        # weights only: a conv bias in front of InstanceNorm3d(affine=False) has an identically-zero gradient (rounding noise in fp32)
        names += [n for n in params if n.startswith("headpose_skin_net.") and n.endswith(".weight") and params[n].grad is not None][-2:]
        names += [n for n in params if n.startswith("model_coarse.XY_gen.") and params[n].grad is not None and params[n].dim() == 4][:1]
        out["h2_%s_grad_names" % tag] = np.array(names)
        for n in names:
            gr = params[n].grad
            out["h2_%s_grad_%s" % (tag, n)] = gr.numpy() if gr.numel() <= 32768 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)].numpy()
            out["h2_%s_gradcks_%s" % (tag, n)] = np.array([gr.double().sum().item(), gr.double().abs().sum().item()])
        n_with = sum(p.grad is not None for p in params.values())
        print("H2", tag, "loss %.6f psnr %.3f" % (loss.item(), psnr), {k: round(v.item(), 6) for k, v in parts.items()}, "params with grad", n_with, "/", len(params))
    # ---------------- H2-ref: the reference SCRIPT's own step --------------------------------------------------------------
    # train_avatar.py cannot be imported (top-level cv2 / lpips / tensorboard, body inside main()), but its per-step statements can
    # be EXECUTED: the lines from `mv_rays = train_batch['mv_rays']` to the learning-rate update (train_avatar.py:108-158) are read
    # from the reference checkout at generation time, dedented and exec'd against the reference Trainer, a real torch.optim.Adam and
    # a batch of this repo's reader (pinned separately against the reference reader, tests/golden/small.npz).  What is stored are
    # numbers only: loss, psnr, the new learning rate and parameter slices AFTER optimizer.step() -- the loss expression, the
    # backward/step/zero_grad order and the decay formula are the reference's text, not a restatement.
    import textwrap
    src = open(os.path.join("/root/reference", "train_avatar.py")).read().split("\n")
    a = next(i for i, ln in enumerate(src) if "mv_rays = train_batch['mv_rays'].to(device)" in ln)
    b = next(i for i, ln in enumerate(src) if 'param_group["lr"] = lr_new' in ln)
    step_code = textwrap.dedent("\n".join(src[a:b + 1]))
    from utils.training_util import mse2psnr as ref_mse2psnr
    cfg = CfgNode(synth.harness_config(perturb=False, noise_std=0.0))
    np.random.seed(7)
    tl = TrainLoader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                     white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    torch.manual_seed(5)
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset)))
    trainer.train()
    optimizer = getattr(torch.optim, cfg.optimizer.type)([{"params": trainer.parameters()}], lr=cfg.optimizer.lr)      # train_avatar.py:71
    ns = {"train_batch": batch, "idx": idx, "device": "cpu", "trainer": trainer, "cfg": cfg, "optimizer": optimizer, "torch": torch,
          "F": torch.nn.functional, "rgb_loss_func": torch.nn.functional.mse_loss, "mse2psnr": ref_mse2psnr, "i": 41}
    for step in range(2):                          # two consecutive steps: the second one sees Adam's state and the decayed rate
        ns["i"] += 1
        exec(step_code, ns)
        out["h2ref_loss_%d" % step] = np.array(ns["loss"].item())
        out["h2ref_psnr_%d" % step] = np.array(ns["psnr"])
        out["h2ref_lr_%d" % step] = np.array(ns["lr_new"])
    out["h2ref_iter"] = np.array(ns["i"])
    after = dict(trainer.named_parameters())
    names = ["model_coarse.layers_xyz.0.weight", "model_coarse.fc_alpha.weight", "model_coarse.fc_rgb.bias", "latent_codes",
             "headpose_skin_net.canonical_Wvolume.final_conv.weight", "model_coarse.XY_gen.conv1.conv.weight"]
    out["h2ref_names"] = np.array(names)
    for n in names:
        t = after[n].detach()
        out["h2ref_after_%s" % n] = t.numpy() if t.numel() <= 32768 else t.reshape(-1)[:: max(1, t.numel() // 4096)].numpy()
    print("H2-ref", [float(out["h2ref_loss_%d" % k]) for k in range(2)], [float(out["h2ref_lr_%d" % k]) for k in range(2)])
    path = os.path.join(REPO, "tests", "golden", "harness.npz")
    np.savez_compressed(path, **out)
    print("harness.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
