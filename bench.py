#!/usr/bin/env python3
"""bench.py -- rendered frames/s of the HAvatar ray-march hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (backend nccl == RCCL).  One "step" = one pass of the hot path over one
frame of synthetic input per rank: 512x512 rays x (64 coarse + 48 fine) MLP queries (BASELINE.json configs[1]);
frames are independent, so N ranks render N frames per step with no data-path collective ("weak" scaling).
Rank 0 prints ONE JSON line.

Other workloads (not what the driver runs; same JSON contract):
  --workload cfg3 [--frames 64]   BASELINE configs[2]: a batch of 64 frames dealt round-robin to the ranks (8 per GPU on 8 GPUs), finished
                                  RGB frames all-gathered round by round over RCCL WHILE the next frame renders
                                  (frames.OverlappedFrameGather); one step = one batch, total work fixed -> "scaling": "strong".
  --workload cfg4                 BASELINE configs[3]: the stage-two HD path, one step = the cfg2 frame (512^2 NeRF volume render) + SWGAN_unet
                                  (512 -> 1024) on its 64 feature channels -> [1,3,1024,1024]; both stages are hipGraphs.
  --workload cfg5                 BASELINE configs[4]: train_avatar.py's optimisation step, 2 frames x 4096 rays x (64 + 48) samples,
                                  forward + backward + Adam as one hipGraph launch, radiance MLP forward/backward on bf16 MFMA.
  --device cpu [--size 16]        plumbing mode for the tests: CPU tensors, gloo, no HIP library; exercises the N > 1 branch without GPUs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 512                                            # (--size overrides, plumbing mode only)
S_C, S_F = 64, 16
Q_PER_RAY = S_C + (S_C + 1) // 2 + S_F                 # 112 MLP queries per ray
FLOP_PER_QUERY = 2 * (176 * 128 + 128 * 128 + 128 * 1 + 128 * 64 + 64 * 3)      # 94 848
FLOP_PER_FRAME = FLOP_PER_QUERY * Q_PER_RAY * H * W     # 2.7848e12
BYTES_PER_FRAME = 600 * H * W + 8388608 + 2097152 + 190992   # compulsory HBM bytes
PEAK_FP32_MFMA = 157.3e12                               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA = 2500e12                                # MI355X_MICROARCH.md: bf16 MFMA, dense
DTYPE = {"half": "f32 emulated as 2 x fp16 split-operand MFMA with fp32 accumulate: 22-bit operands (hi + lo fp16, lo.lo dropped), fp16 exponent range "
                 "guarded on the device (bf16_split_mode = the >= 24-bit-operand mode, exact_f32_mode = true fp32 MFMA, both timed in this line)",
         "split": "f32 (3 x bf16 split-operand MFMA, fp32 accumulate; fp32-sgemm-class results)", "f32": "f32"}
# matrix-core work the kernel actually executes per 32-sample tile: 11 k-chunks x 4 row tiles x 6 products of
# v_mfma_f32_32x32x16_bf16 (32768 FLOP each), or 352 v_mfma_f32_32x32x2_f32 (4096 FLOP each) in the exact-fp32 mode
EXEC_FLOP_PER_TILE = {"half": 132 * 32768, "split": 264 * 32768, "f32": 352 * 4096}       # MFMA instructions per 32-query tile x FLOP each
# feature parking (fp16 cache kernels): the 48 parked tiles of the 80 evaluated per ray block also run fc_rgbFeat on
# the matrix cores, 8 chunks x 2 row tiles x 3 products = 48 more -> 132 + 48 * 48/80 = 160.8 per evaluated tile (= SQ_INSTS_MFMA)
PARK_FLOP_PER_TILE = 48 * 32768
CFG3_NOTE = ("cfg3: one step = a batch of %d frames dealt round-robin to %d rank(s) (%d per rank); each frame is the cfg2 workload below; the "
             "finished RGB frame of round r ([3,512,512] fp32 = 3.1 MB per rank) is all-gathered over RCCL while round r+1 renders")



def emit_line(obj):
    """The ONE JSON line of the contract, as the LAST line of stdout: RCCL (NCCL_DEBUG=VERSION is exported on the GPU boxes) writes its
    version banner through C stdio, which sits in libc's buffer until exit and would land behind a plain print().  Flush libc first."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(obj), flush=True)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def live_pmc_traffic():
    """HBM-side bytes per launch of the march kernel, MEASURED NOW: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: they do
    not fit one pass) over tools/march_once.py, which launches the production kernel on a 512x512 frame and a streaming launch of
    known size; the counters are scaled by the factors that launch calibrates (MI355X_MICROARCH.md, HBM section: on gfx950
    FETCH_SIZE reports half the bytes of a 16 B/lane streaming read, WRITE_SIZE is uncalibrated).  None if rocprofv3 is unusable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    got, cal_meta = {}, None
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("HAV_ABLATE", None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hav_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                sys.executable, os.path.join(ROOT, "tools", "march_once.py")], cwd="/tmp", env=env, timeout=300,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            for ln in r.stdout.decode(errors="replace").splitlines():
                if ln.startswith("{") and "calib_kernel" in ln:
                    cal_meta = json.loads(ln)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    # by full name: in fp16 mode every call also dispatches the range guard's bf16 stand-in, which returns at once
                    name = row["Kernel_Name"].replace("void ", "").split("(")[0].strip()
                    key = "calib" if "fba_vec_kernel" in name else name
                    got.setdefault((key, counter), []).append(float(row["Counter_Value"]))
        except (subprocess.SubprocessError, OSError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    try:
        mean = lambda k: sum(got[k]) / len(got[k])
        mk = cal_meta["march_kernel"]
        rd = mean((mk, "FETCH_SIZE")) * cal_meta["calib_read_bytes"] / mean(("calib", "FETCH_SIZE"))
        wr = mean((mk, "WRITE_SIZE")) * cal_meta["calib_write_bytes"] / mean(("calib", "WRITE_SIZE"))
    except (KeyError, TypeError, ZeroDivisionError):
        return None
    return {"bytes": int(rd + wr), "read": int(rd), "write": int(wr), "kernel": cal_meta["march_kernel"],
            "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/march_once.py in this run, %d launches, "
                      "calibrated on %s (%.0f B per FETCH KB, %.0f B per WRITE KB)" % (
                          len(got[(mk, "FETCH_SIZE")]), cal_meta["calib_kernel"],
                          cal_meta["calib_read_bytes"] / mean(("calib", "FETCH_SIZE")), cal_meta["calib_write_bytes"] / mean(("calib", "WRITE_SIZE")))}


def cpu_baseline(sc, rows, threads):
    """The oracle (C restatement of the reference's algorithm, OpenMP) on `rows` image rows of the same frame."""
    from havatar_amd import synth
    from oracle import oracle
    sub = dict(sc)
    y0 = H // 2 - rows // 2
    r = synth.camera_rays(H, W, y0=y0, y1=y0 + rows)
    sub["rays"] = r[None]
    sub["bg"] = sub["bg"][:, : r.shape[0]]
    oracle.render_rays({**sub, "rays": r[None, :64], "bg": sub["bg"][:, :64]}, S_C, S_F, nthreads=threads)   # warm-up
    t0 = time.perf_counter()
    oracle.render_rays(sub, S_C, S_F, nthreads=threads)
    dt = time.perf_counter() - t0
    return dt * (H / rows), dt


def run_cfg5(args):
    """BASELINE configs[4]: one optimisation step of train_avatar.py (reference train_avatar.py:106-158) -- B = 2 frames x 4096 rays
    (a 64x64 patch each) x (64 coarse + 48 fine) samples = 917 504 radiance-MLP queries, forward + backward + Adam, stratified
    jitter and density noise on -- replayed as ONE hipGraph launch.  The radiance MLP runs forward and backward on hand-written
    bf16-MFMA kernels (hav_mlp_*), recomputing activations in the backward; the rest of the step is DESIGN.md section 7."""
    import tempfile

    import numpy as np
    import torch
    from havatar_amd import synth
    from havatar_amd.dataloader.dataloader import Loader
    from havatar_amd.harness import train
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("cfg5 is a single-GPU workload (the reference trains on one GPU; DP is not part of the north star)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    tmp = tempfile.mkdtemp()
    split = synth.write_dataset(tmp, n_frames=2, img_res=512)
    cfgd = synth.harness_config(render_size=128, gen_size=512, img_res=512, perturb=True, noise_std=0.1, rays=4096)
    cfgd["experiment"]["patch_rgb"] = True            # 64x64 patch = 4096 rays per frame, as the reference trains (dataloader.py:43)
    cfg = CfgNode(cfgd)
    np.random.seed(0)
    torch.manual_seed(0)
    tl = Loader(split_file=split, mode="train", batch_size=2, num_workers=0, down_sample=cfg.dataset.down_sample, options=cfg,
                white_bg=True, shuffle=False)
    idx, batch = next(iter(tl))
    trainer = synth.fill_state_dict(Trainer(cfg, len(tl.dataset))).to(dev).train()
    use_graph = bool(args.graph) and train.graph_training_enabled(dev)
    opt = train.make_optimizer(cfg, trainer, use_graph)
    inp, target, mask = train.step_inputs(idx, batch, dev)
    runner = train.StepRunner(trainer, cfg, opt, torch.nn.functional.mse_loss, graph=use_graph)
    for _ in range(max(args.warmup, 4)):              # 2 eager steps (solver search, optimiser state), capture, first replays
        runner(inp, target, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner(inp, target, mask)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    rays = inp["ray_batch"].shape[0] * inp["ray_batch"].shape[1]
    queries = rays * Q_PER_RAY
    from havatar_amd.native import mlp_train
    info = mlp_train.bench_kernels(trainer.model_coarse, queries, dev)          # HIP events around the MLP kernels alone, same shapes
    flop = 3 * FLOP_PER_QUERY * queries                                        # forward + (data + weight) gradients = 3x forward
    kern_s = info["fwd_ms"] * 1e-3 + info["bwd_ms"] * 1e-3
    res = {"metric": "train_avatar.py optimisation steps/s (2 x 4096 rays x 64+48 samples, fwd + bwd + Adam)", "value": round(1.0 / dt, 3),
           "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 4), "ms_per_step": round(1e3 * dt, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16 (radiance MLP operands, fp32 accumulate and fp32 master weights; encoders / compositing / optimiser fp32)",
           "data": "synthetic",
           "config": {"workload": "cfg5: Trainer.forward(mode=train) + loss (train_avatar.py:121-146 without LPIPS) + backward + Adam, "
                                  "B=2 x 4096 rays, perturb on, radiance_field_noise_std 0.1", "rays": rays, "queries": queries,
                      "hipgraph": use_graph, "loss": round(float(loss), 6), "mlp_mode": info["mode"],
                      "phase_ms": {"mlp_forward_2_passes": round(info["fwd_ms"], 3), "mlp_backward_2_passes": round(info["bwd_ms"], 3)}},
           "roofline": {"bound": "mfma", "achieved": round(flop / kern_s / 1e12, 2), "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                        "frac": round(flop / kern_s / PEAK_BF16_MFMA, 4), "traffic": None, "flop_per_step": flop,
                        "hbm_algorithmic_bytes": info["bytes"], "hbm_achieved_GBps": round(info["bytes"] / kern_s / 1e9, 1),
                        "hbm_frac_of_8TBps": round(info["bytes"] / kern_s / 8e12, 4),
                        "note": "the dominant kernels of the path under training = the radiance MLP forward + backward (hav_mlp_fwd / "
                                "hav_mlp_bwd_data / hav_mlp_bwd_weights); achieved = 3 x 94848 FLOP x queries / their summed time; at 917 504 "
                                "queries the contraction is small (261 GFLOP) and the kernels stream X / dX / activations: both fractions "
                                "are reported"}}
    emit_line(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="image rows timed on the CPU (0 = auto, ~15 s)")
    ap.add_argument("--perturb", type=int, default=1, help="stratified jitter on (reference default for inference)")
    ap.add_argument("--graph", type=int, default=1, help="replay the frame as one hipGraph (0 = eager launches)")
    ap.add_argument("--live-pmc", type=int, default=1, help="measure roofline.traffic in this run (2 rocprofv3 --pmc passes, ~1 min; N=1 only)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2")
    ap.add_argument("--frames", type=int, default=64, help="cfg3: frames per batch (one step = one batch)")
    ap.add_argument("--force-collective", type=int, default=0,
                    help="cfg3 at N=1: create a 1-rank RCCL group and send every finished frame through the side-stream all_gather "
                         "(the exchange step's HIP branch on one GPU)")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda")
    ap.add_argument("--size", type=int, default=512, help="frame edge in pixels (plumbing mode)")
    args = ap.parse_args()
    global H, W, FLOP_PER_FRAME, BYTES_PER_FRAME
    H = W = args.size
    FLOP_PER_FRAME = FLOP_PER_QUERY * Q_PER_RAY * H * W
    BYTES_PER_FRAME = 600 * H * W + 8388608 + 2097152 + 190992
    if args.workload == "cfg5":
        return run_cfg5(args)
    cpu = args.device == "cpu"
    if cpu:
        args.graph, args.live_pmc, args.no_cpu_baseline = 0, 0, True
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # one MIOpen user database / kernel cache per rank: N processes selecting solvers for the same ~40 convolution shapes at
        # the same time otherwise queue on the locks of one shared sqlite file
        for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
            if var not in os.environ:
                d = os.path.join("/tmp", "havatar_miopen_rank%s" % os.environ.get("LOCAL_RANK", "0"), sub)
                os.makedirs(d, exist_ok=True)
                os.environ[var] = d

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.backends.cudnn.benchmark = True                      # MIOpen: pick the fastest solver per conv shape during warm-up
    if cpu:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1) // 2))
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1 or (args.force_collective and args.workload == "cfg3"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)        # backend "nccl" IS RCCL on ROCm
    sync = (lambda: None) if cpu else torch.cuda.synchronize

    from havatar_amd import synth
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode

    # the reference's Trainer surface, scaled to a 512x512 NeRF frame (BASELINE config 2), key-derived synthetic weights
    cfg = CfgNode.load_yaml(os.path.join(ROOT, "havatar_amd", "config", "hd_base.yml"))
    cfg.models.StyleUnet.inp_size = H
    v = cfg.nerf.validation
    v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = S_C, S_F, bool(args.perturb), 0.0
    torch.manual_seed(0)
    tr = Trainer(cfg, 1)
    tr.requires_grad_(False)
    synth.fill_state_dict(tr)
    tr = tr.to(dev)
    tr.headpose_skin_net.fix_canonical_W()                     # inference: frozen skinning volume (avatarHD_reenactment.py:144)
    sc = synth.scene(8, 8, "primary")                           # CPU-baseline constants
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    front, left, right = [t(a) for a in synth.cond_images()]
    rays = t(synth.camera_rays(H, W))[None]
    bg = torch.ones(1, H * W, 3, device=dev)
    # frame k of the batch has its own head pose; rank r renders frames r, r+N, ...
    poses = [t(synth.frame_pose((rank + world * i) % 64))[None] for i in range(8)]
    perturb = bool(args.perturb)

    data = dict(ray_batch=rays, background_prior=bg, inv_head_T=poses[0], front_render_cond=front, left_render_cond=left,
                right_render_cond=right, mode="validation", fidx=0, render_full_img=True)
    if args.graph:
        from havatar_amd.graph import GraphedForward
        frame = GraphedForward(tr, data)                       # the whole frame = one hipGraph launch (+ the pose copy)

        def render(pose):
            return frame(inv_head_T=pose)
    else:
        def render(pose):
            with torch.no_grad():
                return tr(**{**data, "inv_head_T": pose})

    if args.workload == "cfg3":
        # one step = one batch of --frames frames: this rank renders frames rank, rank + N, ...; the finished RGB frame of round r is
        # all-gathered (RCCL over xGMI) while round r + 1 renders; every rank ends the step holding the whole batch
        from havatar_amd.frames import OverlappedFrameGather
        gather = OverlappedFrameGather(args.frames, (3, H, W), device=dev, force_collective=bool(args.force_collective))
        batch_poses = {k: t(synth.frame_pose(k % 64))[None] for k in range(rank, args.frames, world)}

        def step(i):
            for r in range(gather.rounds):
                k = gather.my_frame(r)
                gather.submit(r, None if k is None else render(batch_poses[k])[0][0, :3])
            return gather.finalize()
        frames_per_step = args.frames
    elif args.workload == "cfg4":
        # stage two on top of the frame (reference: avatarHD_reenactment.py:153-160): SWGAN_unet(styles=[style], condition_img=render[:, 3:])
        from havatar_amd.model.styleUnet import SWGAN_unet
        up = SWGAN_unet(inp_size=H, inp_ch=64, out_ch=3, out_size=2 * H, style_dim=64, n_mlp=4, channel_multiplier=2)
        up.requires_grad_(False)
        up = synth.fill_state_dict(up, seed=2).to(dev).eval()
        style = torch.from_numpy(synth.normal((1, 64), 93)).to(dev)
        if args.graph:
            from havatar_amd.graph import GraphedForward
            with torch.no_grad():
                first = render(poses[0])[0]
            up_g = GraphedForward(lambda condition_img: up(styles=[style], condition_img=condition_img),
                                  {"condition_img": first[:, 3:].contiguous()})

            def upsample(feat):
                return up_g(condition_img=feat)
        else:
            def upsample(feat):
                with torch.no_grad():
                    return up(styles=[style], condition_img=feat)

        def step(i):
            return upsample(render(poses[i % len(poses)])[0][:, 3:])
        frames_per_step = world
    else:
        def step(i):
            return render(poses[i % len(poses)])
        frames_per_step = world

    for i in range(args.warmup):
        step(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fps = frames_per_step * args.steps / dt
    if cpu:
        # plumbing mode: no kernels to profile; report the contract fields and what the collective moved
        if rank == 0:
            emit_line(({"metric": "rendered frames/sec @%d^2, 64 samples/ray" % H, "value": round(fps, 4), "unit": "frames/s",
                              "n_gpus": 0, "ranks": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
                              "scaling": "strong" if args.workload == "cfg3" else "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": args.workload + " (CPU plumbing mode: PyTorch statement of the path, gloo)",
                                                               "frames_per_step": frames_per_step, "size": H,
                                                               "gathered": list(out.shape) if args.workload == "cfg3" else None}}))
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- outside the timed region: per-phase device times (HIP events on the launch stream = torch's current stream) ----
    def timed(fn, n):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a_, b_ in ev:
            a_.record(); fn(); b_.record()
        torch.cuda.synchronize()
        return float(np.median([a_.elapsed_time(b_) for a_, b_ in ev]))

    m = tr._hip_marcher()
    vol = tr.headpose_skin_net.current_volume()
    n_ev = max(3, min(args.steps, 10))
    with torch.no_grad():
        kern_ms = timed(lambda: m.render(rays, bg, poses[0], vol, S_C, S_F, perturb=perturb, coarse_outputs=False), n_ev)
        prep_ms = timed(lambda: m.set_triplane(tr.model_coarse.triPlane_embeddings), n_ev)
        enc_ms = timed(lambda: tr.model_coarse.set_conditional_embedding(
            front_render_cond=front, left_render_cond=left, right_render_cond=right, latents=tr.latent_codes[0:1],
            cond_c=poses[0].view(1, -1)), n_ev)
    rm = m

    MODE = {"f32": "f32", "split": "split", "bf16": "split"}.get(os.environ.get("HAVATAR_MLP", "half"), "half")
    F32 = MODE == "f32"

    def committed_pmc(kernel):
        """Newest committed rocprofv3 PMC summary (profiles/*_pmc.json, written by tools/profile.sh) that holds `kernel`."""
        import glob
        import re
        natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]      # r01_v11 after r01_v9
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=natural, reverse=True):
            try:
                d = json.load(open(f))
                return d[kernel.split("(")[0].strip()], d["calibration"], "profiles/" + os.path.basename(f)
            except (KeyError, OSError, ValueError):
                continue
        return None, None, None

    def pmc_traffic(kernel):
        """HBM-side bytes per launch of `kernel` from the committed summary (FETCH_SIZE and WRITE_SIZE in separate passes, scaled by
        the factors calibrated in the same run on a streaming launch of known size).  None when no profile of this kernel exists."""
        k, cal, src = committed_pmc(kernel)
        if k is None or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
            return None
        return {"bytes": int(k["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KB"] + k["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KB"]),
                "read": int(k["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KB"]), "write": int(k["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KB"]),
                "source": src + " (committed profile, not this run)"}

    def unit_busy(kernel):
        """Busy fraction of the three units that share this kernel, from the committed counters (per launch; 256 CUs = 1024 SIMDs;
        GRBM_GUI_ACTIVE is summed over the 8 XCDs): matrix cores SQ_VALU_MFMA_BUSY_CYCLES / 1024, vector ALU SQ_ACTIVE_INST_VALU
        (quad-cycles) x 4 / 1024, texture addresser (the L1 gather path) TA_TA_BUSY_sum / 256 -- each over the kernel's cycles."""
        k, _, src = committed_pmc(kernel)
        if k is None or "GRBM_GUI_ACTIVE" not in k:
            return None
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0
        u = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
            u["mfma"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 3)
        if "SQ_ACTIVE_INST_VALU" in k:
            u["valu"] = round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc, 3)
        if "TA_TA_BUSY_sum" in k:
            u["ta"] = round(k["TA_TA_BUSY_sum"] / 256.0 / cyc, 3)
        if "SQ_INSTS_MFMA" in k:
            u["mfma_instructions_per_launch"] = int(k["SQ_INSTS_MFMA"])
        u["source"] = src
        return u

    # exact-fp32 arithmetic mode (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain) beside the headline: same frame, same jitter setting
    f32_ms = None
    if rank == 0 and MODE != "f32":
        from havatar_amd import _lib
        from havatar_amd.render import RayMarcher
        mf = RayMarcher(m.nerf_scale, m.nerf_trans, m.skin_scale, m.skin_trans)
        mf.mlp_mode = _lib.HAV_MLP_F32
        with torch.no_grad():
            mf.set_mlp(*[t_.detach() for t_ in tr.model_coarse.mlp_tensors()])
            mf.set_triplane(tr.model_coarse.triPlane_embeddings.detach())
            f32_ms = timed(lambda: mf.render(rays, bg, poses[0], vol, S_C, S_F, perturb=perturb, coarse_outputs=False), 3)
        f32_variant = mf.last_variant

    # the >= 24-bit-operand mode (3 x bf16 per operand, six exact products: HAV_MLP_SPLIT_BF16) beside the headline as well
    bf16_ms = None
    if rank == 0 and MODE == "half":
        from havatar_amd import _lib
        from havatar_amd.render import RayMarcher
        mb = RayMarcher(m.nerf_scale, m.nerf_trans, m.skin_scale, m.skin_trans)
        mb.mlp_mode = _lib.HAV_MLP_SPLIT_BF16
        with torch.no_grad():
            mb.set_mlp(*[t_.detach() for t_ in tr.model_coarse.mlp_tensors()])
            mb.set_triplane(tr.model_coarse.triPlane_embeddings.detach())
            bf16_ms = timed(lambda: mb.render(rays, bg, poses[0], vol, S_C, S_F, perturb=perturb, coarse_outputs=False), 5)
        bf16_variant = mb.last_variant

    if rank == 0:
        kname = rm.variant(S_C, S_F, perturb=perturb, coarse_outputs=False)
        traffic = (live_pmc_traffic() if (args.live_pmc and world == 1) else None) or pmc_traffic(kname)
        busy = unit_busy(kname)
        # field evaluations the kernel actually executes per ray: with the fine-pass cache (variants <.., 1> / <.., 2>)
        # the 32 even coarse samples that the merged fine list repeats are not evaluated again
        cached = kname.endswith((", 1>", ", 2>"))
        q_exec = (S_C + S_F) if cached else Q_PER_RAY
        tiles = H * W * q_exec // 32
        exec_flop = EXEC_FLOP_PER_TILE[MODE] * tiles
        if MODE == "half" and cached:                  # feature parking: fc_rgbFeat on the matrix cores for the 48 parked tiles of a block
            exec_flop += PARK_FLOP_PER_TILE * (H * W // 32) * ((S_C + 1) // 2 + S_F)
        peak = PEAK_FP32_MFMA if F32 else PEAK_BF16_MFMA
        # which unit is actually busiest (committed counters of this variant): the label the fraction below must be read with
        names = {"ta": "ta (texture addresser = the L1 gather path of the 8 tri-plane taps)", "mfma": "mfma", "valu": "valu"}
        busiest = max((k for k in ("ta", "mfma", "valu") if busy and k in busy), key=lambda k: busy[k], default=None)
        # the unit that limits the kernel gets its own roofline: bytes the 8 tri-plane taps move into VGPRs per launch (128 x 16 B per lane
        # and evaluated tile: the byte count is fixed by the algebra of DESIGN.md 3.3) against the texture path's 64 B/clk/CU
        tap_bytes = 128 * 16 * 64 * tiles
        kk, _, _src = committed_pmc(kname)
        clk = (kk["GRBM_GUI_ACTIVE"] / 8.0 / (kern_ms * 1e-3)) if (kk and "GRBM_GUI_ACTIVE" in kk) else None       # effective shader clock
        clk_used = clk if (clk and 1.0e9 < clk < 2.6e9) else 2.4e9
        ta_peak = 64.0 * 256 * clk_used
        ta_roof = {"what": "tri-plane tap bytes delivered to VGPRs per launch / (64 B/clk/CU x 256 CUs x clock)", "bytes_per_launch": tap_bytes,
                   "achieved_TBps": round(tap_bytes / (kern_ms * 1e-3) / 1e12, 2), "peak_TBps": round(ta_peak / 1e12, 2),
                   "clock_GHz": round(clk_used / 1e9, 3), "clock_source": ("GRBM_GUI_ACTIVE of the committed profile / this run's kernel time" if clk_used == clk else "2.4 GHz maximum clock (no usable profile)"),
                   "frac": round(tap_bytes / (kern_ms * 1e-3) / ta_peak, 4), "busy": (busy or {}).get("ta")}
        cfg4_ms = None
        if args.workload == "cfg4":          # the upsampler alone (its own graph), HIP events around the replay
            with torch.no_grad():
                feat = render(poses[0])[0][:, 3:].contiguous()
                cfg4_ms = timed(lambda: upsample(feat), n_ev)
        res = {
            "metric": ("stage-two HD frames/sec: 512^2 NeRF volume render (64 samples/ray) + SWGAN_unet upsampler to 1024^2" if args.workload == "cfg4"
                       else "rendered frames/sec @512^2, 64 samples/ray"), "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if args.workload == "cfg3" else "weak", "vs_baseline": None, "dtype": DTYPE[MODE],
            "data": "synthetic",
            "config": {"batch": CFG3_NOTE % (args.frames, world, -(-args.frames // world)) if args.workload == "cfg3" else None,
                       "workload": "cfg2: Trainer.forward(render_full_img=True) for one 512x512 frame per GPU per step: tri-plane encoders "
                                   "(P3: 2x StyleGAN_zxc: 3x3 / up-sampling convolutions on split-fp16 MFMA HIP kernels with the block glue fused, stride-2 and 1x1 ones on MIOpen, HIP upfirdn2d/fused_bias_act) -> per-frame plane projection -> fused "
                                   "ray march (P5-P12) over 262144 rays x (64 coarse + 48 fine) = 29.36M radiance-MLP queries -> [1,67,512,512]",
                       "phase_ms": {"encoders_P3": round(enc_ms, 3), "plane_prepare": round(prep_ms, 3), "ray_march_kernel": round(kern_ms, 3),
                                    "encoders_P3_inside_the_graph": round(1e3 * dt / args.steps / (frames_per_step / world) - kern_ms - prep_ms, 3),
                                    "note": "encoders_P3 is timed eagerly on one stream; inside the frame's hipGraph the two encoders run "
                                            "on two streams without launch gaps: step - march - preparation"},
                       "stage_two": ({"upsampler_ms": round(cfg4_ms, 3), "output": [1, 3, 2 * H, 2 * W],
                                      "what": "BASELINE configs[3]: the step is the cfg2 frame below followed by SWGAN_unet(512 -> 1024) on render[:, 3:] "
                                              "(avatarHD_reenactment.py:153-160): its 3x3 / up-sampling convolutions, Haar transforms, upfirdn2d and "
                                              "fused_bias_act are kernels of this library, stride-2 / 1x1 convolutions MIOpen; phase_ms.encoders_P3_inside_the_graph "
                                              "includes the upsampler in this workload"} if args.workload == "cfg4" else None),
                       "rays_per_frame": H * W, "num_coarse": S_C, "num_fine": S_F, "perturb": perturb, "hipgraph": bool(args.graph),
                       "parallelism": ("frames sharded, %d rank(s), one overlapped all_gather of finished frames per round" if args.workload == "cfg3"
                                       else "frames sharded, %d rank(s), no data-path collective") % world,
                       "exchange": (("RCCL all_gather_into_tensor from a side stream, %d-rank group%s" % (world, " (forced at N = 1)" if world == 1 else ""))
                                    if (args.workload == "cfg3" and gather.collective) else None),
                       "kernel": kname},
            "roofline": {"bound": "mfma", "limited_by": busiest, "busiest_unit": names.get(busiest), "unit_busy": busy, "ta": ta_roof,
                         "achieved": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / 1e12, 3), "peak": peak / 1e12,
                         "unit": "TFLOP/s", "frac": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / peak, 4),
                         "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
                         "kernel_ms": round(kern_ms, 3), "flop_per_launch": FLOP_PER_FRAME,
                         "frac_of_fp32_mfma_peak": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / PEAK_FP32_MFMA, 4),
                         "note": "the path's only dense contraction is the radiance MLP, so the roofline is priced in FLOP ('bound': mfma): "
                                 "achieved = ALGORITHMIC fp32 FLOP of the reference network (94848/query x 112 x 262144) / kernel time; peak = the "
                                 "dense peak of the matrix pipe the kernel runs on (16-bit MFMA 2.5 PFLOP/s in the split modes, where one fp32 "
                                 "product costs 3 (fp16) or 6 (bf16) 16-bit products; fp32 MFMA 157.3 TFLOP/s in exact mode).  The matrix cores "
                                 "are NOT what limits the kernel: limited_by / unit_busy (rocprofv3 counters of the committed profile) name the busiest "
                                 "unit -- the texture addresser serving the tri-plane gather, priced in roofline.ta -- and mfma_executed_* counts what "
                                 "the matrix cores really execute (SQ_INSTS_MFMA).  Against the fp32 MFMA peak the algorithmic number is "
                                 "frac_of_fp32_mfma_peak (> 1: 52% of the reference's matrix work is removed by linearity, DESIGN.md 3.3, the "
                                 "fine pass re-uses the even coarse samples, 3.7, and the rest runs on the 16-bit pipe)",
                         "field_evaluations_per_ray": {"reference": Q_PER_RAY, "executed": q_exec},
                         "mfma_executed_TFLOPs": round(exec_flop / (kern_ms * 1e-3) / 1e12, 2),
                         "mfma_executed_frac_of_peak": round(exec_flop / (kern_ms * 1e-3) / peak, 4),
                         "mfma_instructions_per_launch_model": exec_flop // (4096 if F32 else 32768),
                         "hbm_algorithmic_bytes_per_launch": BYTES_PER_FRAME,
                         "hbm_achieved_GBps": round(BYTES_PER_FRAME / (kern_ms * 1e-3) / 1e9, 2), "hbm_frac_of_8TBps": round(BYTES_PER_FRAME / (kern_ms * 1e-3) / 8e12, 5)},
        }
        if f32_ms is not None:
            step_ms = 1e3 * dt / args.steps / (frames_per_step / world)           # per frame of this rank
            res["exact_f32_mode"] = {"kernel": f32_variant, "kernel_ms": round(f32_ms, 3),
                                     "frames_per_s_est": round(1e3 / (step_ms - kern_ms + f32_ms), 2),
                                     "roofline_frac_of_fp32_mfma_peak": round(FLOP_PER_FRAME / (f32_ms * 1e-3) / PEAK_FP32_MFMA, 4),
                                     "note": "HAVATAR_MLP=f32: v_mfma_f32_32x32x2_f32 (an fmaf chain per dot product) instead of the emulated-fp32 "
                                             "split; same frame; estimate = this run's step time with the march kernel time swapped"}
        if bf16_ms is not None:
            step_ms = 1e3 * dt / args.steps / (frames_per_step / world)
            res["bf16_split_mode"] = {"kernel": bf16_variant, "kernel_ms": round(bf16_ms, 3),
                                      "frames_per_s_est": round(1e3 / (step_ms - kern_ms + bf16_ms), 2),
                                      "note": "HAVATAR_MLP=split: every operand = hi + mid + lo bf16 exactly (>= 24 significant bits), the six partial "
                                              "products >= 2^-16 of the leading one; same frame; estimate = this run's step time with the march "
                                              "kernel time swapped"}
        if not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            rows = args.cpu_rows or 4
            est, took = cpu_baseline(sc, rows, threads)
            if not args.cpu_rows and took < 5.0:             # scale the sample to ~15 s of CPU work
                rows = int(min(H, max(4, rows * 15.0 / max(took, 1e-3)))) // 4 * 4
                est, took = cpu_baseline(sc, rows, threads)
            # one core: rows scaled so that it is ~10 s of work for one thread
            rows1 = 2
            est1, took1 = cpu_baseline(sc, rows1, 1)
            if took1 < 5.0:
                rows1 = int(max(2, min(H, rows1 * 10.0 / max(took1, 1e-3))))
                est1, took1 = cpu_baseline(sc, rows1, 1)
            res["cpu_baseline"] = {"value": round(1.0 / est, 5), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "scope": "the ray march only (P5-P12); the GPU step above also runs both tri-plane encoders (P3: 327 GFLOP of fp32 "
                                            "convolutions per frame, 0.9 s on 8 host cores with PyTorch CPU: BASELINE.md section 2), so the whole-frame CPU "
                                            "rate is lower than this figure",
                                   "cpu_model": cpu_model(),
                                   "sample": "%d of %d image rows (%d rays) of the same frame, oracle/hav_oracle.c with OpenMP, "
                                             "%.1f s measured, scaled to a full frame" % (rows, H, rows * W, took),
                                   "one_core": {"value": round(1.0 / est1, 6), "unit": "frames/s", "cores": 1,
                                                "sample": "%d image row(s) (%d rays), 1 thread, %.1f s measured, scaled to a full frame" % (rows1, rows1 * W, took1)}}
        if dist.is_initialized():          # first, so that nothing the communicator prints on its way out lands behind the line
            dist.barrier()
            dist.destroy_process_group()
        emit_line(res)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
