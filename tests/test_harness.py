"""Harness rows H1 / H2: the reenactment CLI and the stage-one training step against tests/golden/harness.npz,
which holds what the REFERENCE modules produce for the same split file, checkpoint and RNG seeds (oracle/gen_golden_harness.py)."""
import os

import numpy as np
import pytest
import torch
import yaml

from havatar_amd import synth
from havatar_amd.dataloader import imgio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HAS_GPU = torch.cuda.is_available()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "harness.npz"))


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("dataset")
    return str(root), synth.write_dataset(str(root), n_frames=2, img_res=128)


def _write_cfg(path, **kw):
    with open(path, "w") as f:
        yaml.safe_dump(synth.harness_config(**kw), f)
    return str(path)


def _checkpoint(path):
    """The checkpoint of the golden run, rebuilt from this repo's modules (weights are a function of the state_dict key)."""
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.model.styleUnet import SWGAN_unet
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config())
    state = torch.get_rng_state()
    tw = synth.fill_state_dict(Trainer(cfg, 3))
    sw = synth.fill_state_dict(SWGAN_unet(inp_size=32, inp_ch=64, out_size=128, out_ch=3, style_dim=64, c_dim=0, n_mlp=4, channel_multiplier=2), seed=1)
    torch.set_rng_state(state)
    torch.save({"nerf_render": synth.zero_noise_weights({k: v.clone() for k, v in tw.state_dict().items()}),
                "latent_codes": tw.state_dict()["latent_codes"].clone(),
                "g_ema": synth.zero_noise_weights({k: v.clone() for k, v in sw.state_dict().items()})}, path)
    return str(path)


def _run_reenact(tmp_path, dataset, device):
    from havatar_amd.harness import reenact
    cfg_path = _write_cfg(tmp_path / "cfg.yml")
    ckpt = _checkpoint(tmp_path / "avatar.pth")
    out = tmp_path / "renders"
    written = reenact.main(["--config", cfg_path, "--ckpt", ckpt, "--savedir", str(out), "--split", dataset[1]], device=device)
    assert [os.path.basename(p) for p in written] == ["0_00.png", "1_00.png"]         # <fidx>_<view:02d>.png under <savedir>/rgb
    return [imgio.imread_rgb(p) for p in written]


def _png_close(img, ref, max_lsb, max_frac):
    d = np.abs(img.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= max_lsb, "max |diff| %d LSB" % d.max()
    assert (d > 0).mean() <= max_frac, "%.4f of the samples differ" % (d > 0).mean()


def test_dataset_reader_layout(dataset):
    """test-mode frame of the SR reader: [1,N,11] rays equal to the SURVEY 8(d) camera, white background, pose = [R^-1; -t]."""
    from havatar_amd.dataloader.dataloaderSR import Loader
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config())
    frames = list(Loader(split_file=dataset[1], mode="test", batch_size=1, options=cfg, down_sample=cfg.dataset.down_sample))
    assert len(frames) == 2
    for n, (idx, b) in enumerate(frames):
        assert int(idx[0]) == n and int(b["fidx"][0]) == n and int(b["vidx"][0]) == 0
        assert tuple(b["mv_rays"].shape) == (1, 1024, 11) and tuple(b["front_render_cond"].shape) == (1, 256, 256, 7)
        assert np.abs(b["mv_rays"][0, :, :8].numpy() - synth.camera_rays(32, 32)).max() <= 2e-7
        assert float(b["mv_rays"][0, :, 8:].min()) == 1.0
        assert np.abs(b["inv_head_T"][0].numpy() - synth.frame_pose(n)).max() <= 1e-6
        m = b["front_render_cond"][0, :, :, 6]
        assert set(np.unique(m.numpy())) <= {0.0, 1.0} and 0.2 < float(m.mean()) < 0.9


def test_imgio_resize_conventions():
    a = (np.arange(8 * 8 * 3).reshape(8, 8, 3) % 251).astype(np.uint8)
    d = imgio.resize_area(a, 0.25)
    assert d.shape == (2, 2, 3) and d[0, 0, 0] == int(np.floor(a[:4, :4, 0].astype(np.float64).mean() + 0.5))
    assert imgio.resize_linear(a, 8) is a
    up = imgio.resize_linear(a[:, :, :1].astype(np.float32), 16)
    assert up.shape == (16, 16, 1) and abs(float(up[0, 0, 0]) - float(a[0, 0, 0])) < 1e-6       # clamped border, half-pixel centres
    assert abs(float(up[1, 1, 0]) - (0.75 * 0.75 * a[0, 0, 0] + 0.75 * 0.25 * (a[0, 1, 0] + a[1, 0, 0]) + 0.0625 * a[1, 1, 0])) < 1e-4
    m = np.zeros((9, 9), np.uint8); m[2:7, 2:7] = 255
    e = imgio.erode_rect(m, 3)
    assert e[3:6, 3:6].min() == 255 and e.sum() == 9 * 255


def test_reenactment_cli_cpu(tmp_path, dataset, gold):
    """H1 on CPU tensors (PyTorch statement of the path): PNGs equal the reference's up to float rounding at the uint8 cut."""
    imgs = _run_reenact(tmp_path, dataset, "cpu")
    for k, img in enumerate(imgs):
        _png_close(img, gold["h1_png_%d" % k], max_lsb=1, max_frac=0.01)


def test_reenactment_cli_cpu_batched_frames(tmp_path, dataset, gold, monkeypatch):
    """Throughput mode of the CLI (HAVATAR_FRAME_BATCH=2: both frames of the split in ONE call of the renderer and of the upsampler):
    same file names, same pixels as the reference's one-frame sequence."""
    monkeypatch.setenv("HAVATAR_FRAME_BATCH", "2")
    monkeypatch.setenv("HAVATAR_WORKERS", "0")
    imgs = _run_reenact(tmp_path, dataset, "cpu")
    for k, img in enumerate(imgs):
        _png_close(img, gold["h1_png_%d" % k], max_lsb=1, max_frac=0.01)


def test_reenactment_style_vector_matches_reference_rng(gold):
    """avatarHD_reenactment.py:147 draws the style AFTER both constructors consumed the seeded RNG: same stream here."""
    from havatar_amd.harness import reenact
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config())
    torch.manual_seed(cfg.experiment.randomseed)
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.model.styleUnet import SWGAN_unet
    Trainer(cfg, 0)
    SWGAN_unet(inp_size=32, inp_ch=64, out_size=128, out_ch=3, style_dim=reenact.su_args.latent, c_dim=0, n_mlp=reenact.su_args.n_mlp,
               channel_multiplier=reenact.su_args.channel_multiplier)
    style = torch.mean(torch.randn(1000, 1, 64), dim=0)
    assert np.array_equal(style.numpy(), gold["h1_style"])


def _train_step(dataset, gold, tag, device):
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    perturb, noise = (False, 0.0) if tag == "det" else (True, 0.1)
    cfg = CfgNode(synth.harness_config(perturb=perturb, noise_std=noise))
    np.random.seed(7)
    tl = Loader(split_file=dataset[1], mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    assert np.array_equal(batch["mv_rays"].numpy(), gold["h2_mv_rays"]) and np.array_equal(batch["mv_rays_gt_color"].numpy(), gold["h2_target"])
    torch.manual_seed(5)              # construction-time random zero_noise[0] of the two encoders: same stream as the golden run
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(device)
    trainer.train()
    inp, target, ray_mask = train.step_inputs(idx, batch, device)
    torch.manual_seed(123)
    loss, parts, psnr = train.training_loss(trainer, cfg, inp, target, ray_mask, torch.nn.functional.mse_loss)
    trainer.model_coarse.triPlane_embeddings.retain_grad()
    loss.backward()
    return trainer, loss, parts, psnr


def _check_step(trainer, loss, parts, psnr, gold, tag, rtol, planes_rtol=None):
    assert abs(loss.item() - float(gold["h2_%s_loss" % tag])) <= rtol * abs(float(gold["h2_%s_loss" % tag]))
    assert abs(psnr - float(gold["h2_%s_psnr" % tag])) <= 1e-2
    for k, v in parts.items():
        g = float(gold["h2_%s_part_%s" % (tag, k)])
        assert abs(v.item() - g) <= rtol * max(abs(g), 1e-3), k
    params = dict(trainer.named_parameters())
    for n in gold["h2_%s_grad_names" % tag]:
        n = str(n)
        gr = params[n].grad.detach().cpu()
        ref = gold["h2_%s_grad_%s" % (tag, n)]
        got = gr.numpy() if gr.numel() <= 32768 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)].numpy()
        scale = np.abs(ref).max()
        assert scale > 0, n
        # the skinning-volume decoder normalises every layer (InstanceNorm3d): its tiny gradients (1e-5) are differences of
        # large cancelling terms and carry ~1e-3 relative fp32 noise; in fp64 this repo and the reference agree to 1e-9 on them
        # (in bf16 the volume's gradient arrives through dX, like the planes': planes_rtol applies, see test_training_step_gpu)
        tol = max(planes_rtol or rtol, 1e-2) if n.startswith("headpose_skin_net.") else rtol
        assert np.abs(got - ref).max() <= tol * scale, (n, np.abs(got - ref).max() / scale)
    gp = trainer.model_coarse.triPlane_embeddings.grad.detach().cpu()
    ref = gold["h2_%s_grad_planes_slice" % tag]
    assert np.abs(gp[:, :, ::8, ::16, ::16].numpy() - ref).max() <= (planes_rtol or rtol) * np.abs(ref).max()
    cks = gold["h2_%s_grad_planes_cks" % tag]
    assert abs(gp.double().abs().sum().item() - cks[1]) <= rtol * cks[1]
    assert sum(p.grad is not None for p in params.values()) == 153            # 153 of 157 parameters receive gradients


@pytest.mark.parametrize("tag", ["det", "rnd"])
def test_training_step_cpu(dataset, gold, tag):
    """H2: loss, its parts and the gradients of one optimisation step vs the reference modules (same RNG stream for `rnd`)."""
    _check_step(*_train_step(dataset, gold, tag, "cpu"), gold, tag, rtol=2e-4)


def test_two_optimisation_steps_match_the_reference_script(dataset, gold):
    """The reference SCRIPT's own per-step statements (train_avatar.py:108-158: inputs, loss expression, backward / step /
    zero_grad order, learning-rate decay), exec'd from its text against the reference Trainer when the fixture was generated
    (oracle/gen_golden_harness.py, H2-ref), against this repo's harness: two consecutive steps from the same initial state --
    losses, PSNR, the decayed learning rate and the parameters AFTER the second Adam update."""
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config(perturb=False, noise_std=0.0))
    np.random.seed(7)
    tl = Loader(split_file=dataset[1], mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    torch.manual_seed(5)
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).train()
    opt = train.make_optimizer(cfg, trainer, graph=False)
    run = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=False)
    i = int(gold["h2ref_iter"]) - 2
    for step in range(2):
        i += 1
        inp, target, ray_mask = train.step_inputs(idx, batch, "cpu")
        loss, parts, psnr = run(inp, target, ray_mask)
        lr_new = train.learning_rate(cfg, i)
        train.set_learning_rate(opt, lr_new)
        ref_loss = float(gold["h2ref_loss_%d" % step])
        assert abs(loss.item() - ref_loss) <= 2e-5 * abs(ref_loss), (step, loss.item(), ref_loss)
        assert abs(psnr - float(gold["h2ref_psnr_%d" % step])) <= 1e-3
        assert abs(lr_new - float(gold["h2ref_lr_%d" % step])) <= 1e-12
    after = dict(trainer.named_parameters())
    for n in gold["h2ref_names"]:
        n = str(n)
        t = after[n].detach()
        got = t.numpy() if t.numel() <= 32768 else t.reshape(-1)[:: max(1, t.numel() // 4096)].numpy()
        ref = gold["h2ref_after_%s" % n]
        # Adam's first steps move every weight by ~lr * sign(gradient) whatever the gradient's size, so an entry whose gradient is
        # rounding noise (|g| ~ 1e-12) may go the other way: the bulk must agree to a fraction of the step, stragglers are bounded by 2 lr per step
        diff = np.abs(got - ref) / cfg.optimizer.lr
        assert np.median(diff) <= 1e-2 and np.mean(diff > 0.05) <= 0.03 and diff.max() <= 4.1, (n, np.median(diff), np.mean(diff > 0.05), diff.max())


def test_training_cli_runs_and_resumes(tmp_path, dataset, monkeypatch):
    """train_avatar counterpart end to end on CPU: 2 steps, checkpoint with the reference's keys, resume from it."""
    from havatar_amd.harness import train
    monkeypatch.setenv("HAVATAR_PRETRAIN_WC", "2")
    monkeypatch.setenv("HAVATAR_WORKERS", "0")
    cfg_path = _write_cfg(tmp_path / "cfg.yml", perturb=True, noise_std=0.1)
    log = tmp_path / "log"
    last = train.main(["--logdir", str(log), "--datadir", dataset[0], "--config", cfg_path, "--max-steps", "1"], device="cpu")
    assert last == 0
    ck = torch.load(log / "checkpoint00000.ckpt", map_location="cpu", weights_only=False)
    assert set(ck) == {"iter", "optimizer_state_dict", "loss", "psnr", "trainer_state_dict"} and ck["iter"] == 0
    assert os.path.exists(log / "vis_motionWeightVol00000.obj") and os.path.exists(log / "val_00000.png")
    last = train.main(["--logdir", str(log), "--datadir", dataset[0], "--config", cfg_path, "--ckpt", str(log / "checkpoint00000.ckpt"),
                       "--max-steps", "1"], device="cpu")
    assert last == 1
    assert train.learning_rate(type("C", (), {"optimizer": type("O", (), {"lr": 5e-4}), "scheduler": type("S", (), {"lr_decay": 250, "lr_decay_factor": 0.1})}), 250000) == pytest.approx(5e-5)


@pytest.mark.gpu
def test_training_cli_gpu_graph_mode_runs_validates_and_resumes(tmp_path, dataset, monkeypatch, capsys):
    """train_avatar counterpart on the device with the step as one hipGraph launch: 5 steps (2 eager, capture, replays) with a
    validation render through the fused inference kernel and a checkpoint, then a resume that captures again."""
    from havatar_amd.harness import train
    monkeypatch.setenv("HAVATAR_PRETRAIN_WC", "2")
    monkeypatch.setenv("HAVATAR_WORKERS", "0")
    cfg_path = _write_cfg(tmp_path / "cfg.yml", perturb=True, noise_std=0.1)
    log = tmp_path / "log"
    last = train.main(["--logdir", str(log), "--datadir", dataset[0], "--config", cfg_path, "--max-steps", "5"], device="cuda")
    assert last == 4
    ck = torch.load(log / "checkpoint00000.ckpt", map_location="cpu", weights_only=False)
    assert set(ck) == {"iter", "optimizer_state_dict", "loss", "psnr", "trainer_state_dict"} and np.isfinite(float(ck["loss"]))
    assert all(isinstance(g["lr"], float) for g in ck["optimizer_state_dict"]["param_groups"])     # reference-compatible: plain floats
    last = train.main(["--logdir", str(log), "--datadir", dataset[0], "--config", cfg_path, "--ckpt", str(log / "checkpoint00000.ckpt"),
                       "--max-steps", "4"], device="cuda")
    assert last == 4
    out = capsys.readouterr().out
    assert "Validation loss" in out and "nan" not in out.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("restored_lr", ["float", "cpu_tensor"])
def test_graph_mode_learning_rate_reaches_the_replayed_step_after_resume(restored_lr):
    """A resumed run's optimiser state holds the rate as a float (this harness, the reference) or as a CPU tensor (graph-mode
    checkpoints of the previous revision, loaded with map_location="cpu").  StepRunner must turn either into a DEVICE tensor before
    capture: a CPU 0-dim rate is baked into the captured Adam update as a constant and the exponential decay is lost.  Checked on
    the parameter trajectory: with the rate set to 0 a replayed step must not move the weights; with it restored it must."""
    from havatar_amd.harness import train
    from havatar_amd.utils.cfgnode import CfgNode
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = torch.nn.Linear(8, 8).to(dev)

    class _T(torch.nn.Module):                   # the smallest "trainer" StepRunner's graph path accepts
        def __init__(self):
            super().__init__()
            self.net = net
    cfg = CfgNode({"optimizer": {"type": "Adam", "lr": 1e-2}})
    tr = _T()
    opt = train.make_optimizer(cfg, tr, graph=True)
    sd = opt.state_dict()
    for g in sd["param_groups"]:
        g["lr"] = 1e-2 if restored_lr == "float" else torch.tensor(1e-2)
        g["capturable"] = False                    # what an eager / reference checkpoint restores
    opt.load_state_dict(sd)
    runner = train.StepRunner(tr, cfg, opt, torch.nn.functional.mse_loss, graph=True)
    assert all(torch.is_tensor(g["lr"]) and g["lr"].is_cuda for g in opt.param_groups)
    from havatar_amd.graph import GraphedTrainStep
    x = torch.randn(4, 8, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.current_stream().wait_stream(side)
    g = GraphedTrainStep(lambda x: (net(x).square().mean(), {}), opt, {"x": x})
    w0 = net.weight.detach().clone()
    g(x=x)
    torch.cuda.synchronize()
    moved = (net.weight.detach() - w0).abs().max().item()
    assert moved > 1e-4
    train.set_learning_rate(opt, 0.0)
    w1 = net.weight.detach().clone()
    g(x=x)
    torch.cuda.synchronize()
    assert torch.equal(net.weight.detach(), w1), "the replayed update ignored the new learning rate"
    train.set_learning_rate(opt, 1e-2)
    g(x=x)
    torch.cuda.synchronize()
    assert not torch.equal(net.weight.detach(), w1)


@pytest.mark.gpu
def test_reenactment_cli_gpu(tmp_path, dataset, gold):
    """H1 through the fused HIP renderer + hipGraph + MIOpen encoders: float outputs within the path's tolerance of the
    reference's, PNGs within 2 LSB."""
    from havatar_amd.dataloader.dataloaderSR import Loader
    from havatar_amd.harness import reenact
    from havatar_amd.utils.cfgnode import CfgNode
    imgs = _run_reenact(tmp_path, dataset, "cuda")
    for k, img in enumerate(imgs):
        _png_close(img, gold["h1_png_%d" % k], max_lsb=2, max_frac=0.05)
    cfg = CfgNode(synth.harness_config())
    ck = torch.load(tmp_path / "avatar.pth", map_location="cpu")
    nerf_render, img_trans = reenact.build_models(cfg, ck, torch.device("cuda"))
    style = torch.from_numpy(gold["h1_style"]).cuda()
    with torch.no_grad():
        for idx, b in Loader(split_file=dataset[1], mode="test", batch_size=1, options=cfg, down_sample=cfg.dataset.down_sample):
            k = int(b["fidx"][0])
            render, mask, _ = nerf_render(**reenact.frame_inputs(idx, b, "cuda"))
            assert (render.cpu().numpy() - gold["h1_render_%d" % k]).__abs__().max() <= 1e-3
            assert (mask.cpu().numpy() - gold["h1_mask_%d" % k]).__abs__().max() <= 1e-3
            gen = img_trans(styles=[style], condition_img=render[:, 3:])
            assert (gen.cpu().numpy() - gold["h1_gen_%d" % k]).__abs__().max() <= 5e-3


@pytest.mark.gpu
def test_reenactment_cli_gpu_batched_frames(tmp_path, dataset, gold, monkeypatch):
    """H1 in throughput mode on the device (HAVATAR_FRAME_BATCH=2: one hipGraph replay per stage for both frames; device-side ray
    generation for a batch): the PNGs of the one-frame sequence, within the same 2 LSB of the reference's."""
    monkeypatch.setenv("HAVATAR_FRAME_BATCH", "2")
    imgs = _run_reenact(tmp_path, dataset, "cuda")
    for k, img in enumerate(imgs):
        _png_close(img, gold["h1_png_%d" % k], max_lsb=2, max_frac=0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("mlp,rtol", [("torch", 2e-3), ("bf16", 2e-2)])
def test_training_step_gpu(dataset, gold, monkeypatch, mlp, rtol):
    """H2 on the device against the reference's autograd (loss, its parts, gradients of MLP / planes / latent codes / volume decoder /
    encoder convs): with the fp32 nn.Linear statement of the radiance MLP at 2e-3, and with the bf16-MFMA kernels of BASELINE
    config 5 (hav_mlp_train_*, the default) at the relaxed 2e-2 SURVEY 8(a) H2 states for the bf16 build."""
    monkeypatch.setenv("HAVATAR_TRAIN_MLP", mlp)
    # gradients that reach their tensor through dX (per-texel plane gradients, the skinning volume) sum over few queries: in bf16 a
    # relu unit within ~1e-3 of zero switches (tests/test_mlp_train_gpu.py) and the entries its query touches move by a few per cent
    # of the largest one; every gradient that sums over all queries (MLP, latent codes, encoder convolutions) is held to 2e-2
    _check_step(*_train_step(dataset, gold, "det", "cuda"), gold, "det", rtol=rtol, planes_rtol=6e-2 if mlp == "bf16" else None)


@pytest.mark.gpu
def test_training_step_gpu_fused_field_ops_equal_the_aten_statement(dataset, gold, monkeypatch):
    """The same jittered, noisy step (same device RNG stream) through hav_field_inputs / hav_composite and through the ATen
    statement of the march: loss, parts and every gradient agree to fp32 noise."""
    runs = []
    monkeypatch.setenv("HAVATAR_TRAIN_MLP", "torch")          # isolate the field / compositing kernels: the MLP is fp32 nn.Linear on both sides
    for flag in ("0", "1"):
        monkeypatch.setenv("HAVATAR_HIP_TRAIN", flag)
        trainer, loss, parts, _ = _train_step(dataset, gold, "rnd", "cuda")
        grads = {n: p.grad.detach().clone() for n, p in trainer.named_parameters() if p.grad is not None}
        grads["planes"] = trainer.model_coarse.triPlane_embeddings.grad.detach().clone()
        runs.append((loss.item(), {k: v.item() for k, v in parts.items()}, grads))
    (l0, p0, g0), (l1, p1, g1) = runs
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    for k in p0:
        assert abs(p0[k] - p1[k]) <= 1e-5 * max(abs(p0[k]), 1e-3), k
    assert g0.keys() == g1.keys() and len(g0) == 154
    for n in g0:
        if n.startswith("headpose_skin_net.canonical_Wvolume.filters") and n.endswith(".bias"):
            continue                                   # a bias in front of InstanceNorm3d: its gradient is zero up to rounding
        scale = g0[n].abs().max().item()
        # the fine pass amplifies coarse rounding ~100x (SURVEY B-11) and a noisy density within rounding of 0 flips its relu:
        # gradients of the two statements agree to ~1e-2 of their scale, the loss to 1e-5
        tol = 2e-2
        assert (g0[n] - g1[n]).abs().max().item() <= tol * scale + 1e-12, (n, (g0[n] - g1[n]).abs().max().item() / max(scale, 1e-30))


@pytest.mark.gpu
def test_training_step_gpu_resampling_kernel_equals_the_aten_statement(dataset, gold, monkeypatch):
    """The same jittered, noisy step (same host / device RNG streams: hav_resample_depths takes the draw sample_pdf would make) with the
    importance resampling as one launch and as the ATen statement of model/nerf_trainer.py:166-170: loss and parts agree to 1e-5, gradients
    to the 2e-2 two statements of the coarse-to-fine hand-over agree to (the fine pass amplifies rounding of the CDF, SURVEY B-11)."""
    from havatar_amd.native import train_ops
    runs, calls = [], []
    real = train_ops.resample_depths
    monkeypatch.setattr(train_ops, "resample_depths", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setenv("HAVATAR_TRAIN_MLP", "torch")
    for flag in ("aten", "hip"):
        monkeypatch.setenv("HAVATAR_RESAMPLE", flag)
        trainer, loss, parts, _ = _train_step(dataset, gold, "rnd", "cuda")
        grads = {n: p.grad.detach().clone() for n, p in trainer.named_parameters() if p.grad is not None}
        runs.append((loss.item(), {k: v.item() for k, v in parts.items()}, grads, len(calls)))
    (l0, p0, g0, c0), (l1, p1, g1, c1) = runs
    assert c0 == 0 and c1 >= 1                       # the kernel is what ran in the second step
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    for k in p0:
        assert abs(p0[k] - p1[k]) <= 1e-5 * max(abs(p0[k]), 1e-3), k
    for n in g0:
        if n.startswith("headpose_skin_net.canonical_Wvolume.filters") and n.endswith(".bias"):
            continue
        scale = g0[n].abs().max().item()
        assert (g0[n] - g1[n]).abs().max().item() <= 2e-2 * scale + 1e-12, n


@pytest.mark.gpu
def test_training_steps_as_one_hipgraph_launch_follow_the_eager_trajectory(dataset, gold, monkeypatch):
    """graph.GraphedTrainStep: forward + backward + Adam of a deterministic-depth step replayed as one graph.  Six steps from the
    same initial state, eager vs (2 eager + 4 replayed): the loss curves agree, the parameters end up equal, and the
    weight-derived caches of the inference path see the replayed updates (weights_epoch)."""
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    # fp32 MLP: the scatter kernels use float atomics, and with bf16 operands a relu unit switched by that last-bit noise moves the
    # trajectory by more than the 1 % this test allows; the bf16 kernels under capture are covered by the CLI test and bench.py cfg5
    monkeypatch.setenv("HAVATAR_TRAIN_MLP", "torch")
    cfg = CfgNode(synth.harness_config(perturb=False, noise_std=0.0))
    np.random.seed(7)
    tl = Loader(split_file=dataset[1], mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    curves, finals, renders = [], [], []
    for graph in (False, True):
        torch.manual_seed(5)
        trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to("cuda").train()
        opt = train.make_optimizer(cfg, trainer, graph)
        run = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=graph)
        init = {n: p.detach().clone() for n, p in trainer.named_parameters()}
        inp, target, mask = train.step_inputs(idx, batch, "cuda")
        val = {"mode": "validation", "fidx": None, "render_full_img": False, "ray_batch": inp["ray_batch"][:1, :256].contiguous(),
               "background_prior": inp["background_prior"][:1, :256].contiguous(), "inv_head_T": inp["inv_head_T"][:1],
               **{k: inp[k][:1] for k in ("front_render_cond", "left_render_cond", "right_render_cond")}}
        losses = []
        for k in range(6):
            loss, parts, psnr = run(inp, target, mask)
            losses.append(loss.item())
            train.set_learning_rate(opt, 5e-4 * 0.9 ** (k + 1))
            if k == 3:                                 # an inference render in the middle: must use the weights of step 3 ...
                trainer.eval()
                with torch.no_grad():
                    renders.append(trainer(**val)[4].clone())
                trainer.train()
        trainer.eval()
        with torch.no_grad():
            renders.append(trainer(**val)[4].clone())     # ... and this one those of step 5
        assert (run.graphed is not None) == graph
        curves.append(losses)
        finals.append({n: p.detach().clone() for n, p in trainer.named_parameters()})
    assert curves[0][0] > curves[0][-1]
    # float atomics in the scatter kernels make every run a little different and six Adam steps amplify it: the bounds below are
    # several times the spread seen between repeated runs, and far below what a stale-weights or wrong-learning-rate bug produces
    print("loss curves", curves)
    for a, b in zip(*curves):
        assert abs(a - b) <= 1e-2 * abs(a), curves
    # Adam moves every element by ~lr per step whatever the size of its gradient, so elements whose gradient is rounding noise end
    # up anywhere within 6 lr of each other; what must agree is the bulk of the movement
    moved = sum((finals[0][n] - init[n]).abs().sum().item() for n in init)
    apart = sum((finals[0][n] - finals[1][n]).abs().sum().item() for n in init)
    print("moved", moved, "apart", apart, "render diffs", (renders[0] - renders[2]).abs().max().item(), (renders[1] - renders[3]).abs().max().item(),
          "render change", (renders[2] - renders[3]).abs().max().item())
    assert moved > 0 and apart <= 0.3 * moved, (apart, moved)
    assert max((finals[0][n] - finals[1][n]).abs().max().item() for n in init) <= 6 * 5e-4 * 1.01
    # the inference renders saw the replayed updates: eager and replayed renders of the SAME step agree (max: within the atomics
    # noise six Adam steps amplify; mean: much better than the renders of step 3 and step 5 differ)
    change = (renders[0] - renders[1]).abs().mean().item()
    print("mean render change", change, "mean same-step diffs", (renders[0] - renders[2]).abs().mean().item(), (renders[1] - renders[3]).abs().mean().item())
    assert change > 1e-6
    for a, b in ((renders[0], renders[2]), (renders[1], renders[3])):
        assert (a - b).abs().max().item() <= 6e-2 and (a - b).abs().mean().item() <= 0.5 * change, ((a - b).abs().mean().item(), change)


@pytest.mark.gpu
def test_graphed_training_step_is_a_single_chain_and_replays_in_order(dataset, monkeypatch, recwarn):
    """The captured optimisation step must not fork: GraphedTrainStep captures on the stream the eager warm-up steps ran on, so the
    parameters' AccumulateGrad nodes (bound to the stream they were created on) add no side branches.  With side branches, replays on
    this stack ran consecutive kernels of the main chain out of order (round 4: NaN losses in 13 % of the CLI test's runs).
    Tripwires: (1) PyTorch's "AccumulateGrad node's stream does not match" warning does not appear; (2) the losses of 2 x 12 steps stay
    finite; (3) is-finite flags of every operand / result of the convolution wrappers, traced INSIDE the captured graph
    (native/conv.py::_trace) -- a kernel that reads a half-written tensor shows up there long before the loss does: NO flag may be raised.
    Round 4 tolerated one flagged replay here; round 5 root-caused that residue to the ROCm runtime (replays of a graph return stale results
    once a reduction kernel was launched eagerly in between: tools/repro_graph_reduce.py) and havatar_amd switches the
    responsible launch path off at import (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): 0 flagged replays of 1 460 in profiles/r05_graph_anomaly_root_cause.txt."""
    import havatar_amd
    assert havatar_amd.hipgraph_replays_safe(), "the test process must run with %s=0" % havatar_amd.HIPGRAPH_PACKET_CAPTURE_ENV
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.native import conv
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config(perturb=True, noise_std=0.1))
    np.random.seed(3)
    tl = Loader(split_file=dataset[1], mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    monkeypatch.setattr(conv, "_NAN_TRACE", [])
    flagged = []
    for run_no in range(2):                        # (the second build of everything in one process is where the flake showed up)
        torch.manual_seed(11 + run_no)
        trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to("cuda").train()
        opt = train.make_optimizer(cfg, trainer, True)
        run = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=True)
        inp, target, mask = train.step_inputs(idx, batch, "cuda")
        for k in range(12):
            loss, _, _ = run(inp, target, mask)
            train.set_learning_rate(opt, 5e-4)
            assert np.isfinite(loss.item()), (run_no, k)
            if run.graphed is not None:
                flags = list(conv._NAN_TRACE)
                assert len(flags) > 50, len(flags)          # the captured step's convolutions are in the trace
                bad = [n for n, f in flags if not bool(f)]
                if bad:
                    flagged.append((run_no, k, bad[:5]))
            if k % 4 == 3:                                  # an eager inference render between replays, as train.main() does
                trainer.eval()
                with torch.no_grad():
                    trainer(mode="validation", fidx=None, render_full_img=False, ray_batch=inp["ray_batch"][:1, :256].contiguous(),
                            background_prior=inp["background_prior"][:1, :256].contiguous(), inv_head_T=inp["inv_head_T"][:1],
                            **{kk: inp[kk][:1] for kk in ("front_render_cond", "left_render_cond", "right_render_cond")})
                trainer.train()
        assert run.graphed is not None
    assert not flagged, flagged
    assert not [w for w in recwarn.list if "AccumulateGrad node's stream does not match" in str(w.message)]


@pytest.mark.gpu
def test_deterministic_switch_makes_the_optimisation_step_bit_reproducible(dataset, monkeypatch):
    """HAVATAR_DETERMINISTIC=1 (VERDICT r4 #4e, weak #11): the same step -- same weights, batch, seeds; jitter and density noise on -- three
    times: the loss and every one of the 153 parameter gradients have the same bits each time.  (Without the switch the float atomics of
    the field-input scatter and MIOpen's default solver for the 16^2 convolutions make 141 of them differ: tools/step_determinism.py.)
    Reference statement: train_avatar.py:121-158."""
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    monkeypatch.setenv("HAVATAR_DETERMINISTIC", "1")
    cfg = CfgNode(synth.harness_config(perturb=True, noise_std=0.1))
    np.random.seed(3)
    tl = Loader(split_file=dataset[1], mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    torch.manual_seed(11)
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to("cuda").train()
    inp, target, mask = train.step_inputs(idx, batch, "cuda")
    train.enable_determinism(True)
    try:
        runs = []
        for k in range(3):
            torch.manual_seed(1234)
            torch.cuda.manual_seed(1234)
            for p in trainer.parameters():
                p.grad = None
            loss, _, _ = train.training_loss(trainer, cfg, inp, target, mask, torch.nn.functional.mse_loss, None)
            loss.backward()
            torch.cuda.synchronize()
            runs.append(dict({n: p.grad.detach().clone() for n, p in trainer.named_parameters() if p.grad is not None}, loss=loss.detach().clone()))
        assert len(runs[0]) == 154
        for other in runs[1:]:
            differ = [n for n in runs[0] if not torch.equal(runs[0][n], other[n])]
            assert not differ, differ[:8]
    finally:
        train.enable_determinism(False)


@pytest.mark.gpu
def test_hipgraph_replays_survive_eager_reductions_with_the_runtime_workaround():
    """The ROCm runtime fault behind round 4's residue, outside this package (tools/repro_graph_reduce.py: PyTorch only): a captured graph of
    element-wise ops, is-finite reductions and sums of ones, replayed 60 times with ONE eager torch.sum every fourth replay.  With the
    setting havatar_amd applies at import (in-process, after `import torch`) every replay is right.  The same script without it is run for the
    record only: the fault belongs to the runtime (56 of 60 replays wrong on ROCm 7.0.2), a fixed runtime makes that line read 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(**env):
        e = dict(os.environ, EAGER="sum", **env)
        e.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "repro_graph_reduce.py")], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("repro_graph_reduce:")][-1]
    fixed = run(WORKAROUND="inprocess")
    print(fixed)
    print("without the workaround:", run())
    assert " 0 of 60 replays" in fixed, fixed


def test_reenactment_cli_shards_frames_across_ranks(tmp_path, dataset, gold, monkeypatch):
    """config 3 plumbing (frames sharded over ranks, no data-path collective): with WORLD_SIZE=2 each rank renders its own frames
    and writes the same files a single process would."""
    from havatar_amd.harness import reenact
    cfg_path = _write_cfg(tmp_path / "cfg.yml")
    ckpt = _checkpoint(tmp_path / "avatar.pth")
    monkeypatch.setenv("HAVATAR_WORKERS", "0")
    names = []
    for rank in (0, 1):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "2")
        written = reenact.main(["--config", cfg_path, "--ckpt", ckpt, "--savedir", str(tmp_path / "out"), "--split", dataset[1]], device="cpu")
        assert [os.path.basename(p) for p in written] == ["%d_00.png" % rank]
        names += written
    for k, p in enumerate(names):
        _png_close(imgio.imread_rgb(p), gold["h1_png_%d" % k], max_lsb=1, max_frac=0.01)
