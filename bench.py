#!/usr/bin/env python3
"""bench.py -- rendered frames/s of the HAvatar ray-march hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (backend nccl == RCCL).  One "step" = one pass of the hot path over one
frame of synthetic input per rank: 512x512 rays x (64 coarse + 48 fine) MLP queries (BASELINE.json configs[1]);
frames are independent, so N ranks render N frames per step with no data-path collective ("weak" scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 512
S_C, S_F = 64, 16
Q_PER_RAY = S_C + (S_C + 1) // 2 + S_F                 # 112 MLP queries per ray
FLOP_PER_QUERY = 2 * (176 * 128 + 128 * 128 + 128 * 1 + 128 * 64 + 64 * 3)      # 94 848
FLOP_PER_FRAME = FLOP_PER_QUERY * Q_PER_RAY * H * W     # 2.7848e12
BYTES_PER_FRAME = 600 * H * W + 8388608 + 2097152 + 190992   # compulsory HBM bytes
PEAK_FP32_MFMA = 157.3e12                               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA = 2500e12                                # MI355X_MICROARCH.md: bf16 MFMA, dense
DTYPE = {"half": "f32 (2 x fp16 split-operand MFMA, fp32 accumulate; fp32-sgemm-class results)",
         "split": "f32 (3 x bf16 split-operand MFMA, fp32 accumulate; fp32-sgemm-class results)", "f32": "f32"}
# matrix-core work the kernel actually executes per 32-sample tile: 11 k-chunks x 4 row tiles x 6 products of
# v_mfma_f32_32x32x16_bf16 (32768 FLOP each), or 352 v_mfma_f32_32x32x2_f32 (4096 FLOP each) in the exact-fp32 mode
EXEC_FLOP_PER_TILE = {"half": 132 * 32768, "split": 264 * 32768, "f32": 352 * 4096}       # MFMA instructions per 32-query tile x FLOP each


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="image rows timed on the CPU (0 = auto, ~15 s)")
    ap.add_argument("--perturb", type=int, default=1, help="stratified jitter on (reference default for inference)")
    ap.add_argument("--graph", type=int, default=1, help="replay the frame as one hipGraph (0 = eager launches)")
    args = ap.parse_args()
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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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

        def step(i):
            return frame(inv_head_T=poses[i % len(poses)])
    else:
        def step(i):
            with torch.no_grad():
                return tr(**{**data, "inv_head_T": poses[i % len(poses)]})

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fps = world * args.steps / dt

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

    def pmc_traffic(kernel):
        """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_pmc.json, written by
        tools/profile.sh: FETCH_SIZE and WRITE_SIZE in separate passes, scaled by the factors calibrated in the same run on a
        streaming launch of known size -- MI355X_MICROARCH.md, HBM section).  None when no profile of this kernel is committed."""
        import glob
        import re
        natural = lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))]      # r01_v11 after r01_v9
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc.json")), key=natural, reverse=True):
            try:
                d = json.load(open(f))
                k, cal = d[kernel.split("(")[0].strip()], d["calibration"]
                return {"bytes": int(k["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KB"] + k["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KB"]),
                        "read": int(k["FETCH_SIZE"] * cal["read_bytes_per_FETCH_KB"]), "write": int(k["WRITE_SIZE"] * cal["write_bytes_per_WRITE_KB"]),
                        "source": "profiles/" + os.path.basename(f)}
            except (KeyError, OSError, ValueError):
                continue
        return None

    if rank == 0:
        kname = rm.variant(S_C, S_F, perturb=perturb, coarse_outputs=False)
        traffic = pmc_traffic(kname)
        # field evaluations the kernel actually executes per ray: with the fine-pass cache (variants <.., 1> / <.., 2>)
        # the 32 even coarse samples that the merged fine list repeats are not evaluated again
        q_exec = (S_C + S_F) if kname.endswith((", 1>", ", 2>")) else Q_PER_RAY
        res = {
            "metric": "rendered frames/sec @512^2, 64 samples/ray", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[MODE], "data": "synthetic",
            "config": {"workload": "cfg2: Trainer.forward(render_full_img=True) for one 512x512 frame per GPU per step: tri-plane encoders "
                                   "(P3: 2x StyleGAN_zxc, MIOpen convs + HIP upfirdn2d/fused_bias_act) -> per-frame plane projection -> fused "
                                   "ray march (P5-P12) over 262144 rays x (64 coarse + 48 fine) = 29.36M radiance-MLP queries -> [1,67,512,512]",
                       "phase_ms": {"encoders_P3": round(enc_ms, 3), "plane_prepare": round(prep_ms, 3), "ray_march_kernel": round(kern_ms, 3)},
                       "rays_per_frame": H * W, "num_coarse": S_C, "num_fine": S_F, "perturb": perturb, "hipgraph": bool(args.graph),
                       "parallelism": "frames sharded, %d rank(s), no data-path collective" % world,
                       "kernel": rm.variant(S_C, S_F, perturb=perturb, coarse_outputs=False)},
            "roofline": {"bound": "mfma", "achieved": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / 1e12, 3),
                         "peak": (PEAK_FP32_MFMA if F32 else PEAK_BF16_MFMA) / 1e12,
                         "unit": "TFLOP/s", "frac": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / (PEAK_FP32_MFMA if F32 else PEAK_BF16_MFMA), 4),
                         "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
                         "kernel_ms": round(kern_ms, 3), "flop_per_launch": FLOP_PER_FRAME,
                         "frac_of_fp32_mfma_peak": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / PEAK_FP32_MFMA, 4),
                         "note": "achieved = ALGORITHMIC fp32 FLOP of the reference network (94848/query) / kernel time; peak = the dense peak "
                                 "of the matrix pipe the kernel runs on (16-bit MFMA 2.5 PFLOP/s in the split modes, where one fp32 product "
                                 "costs 3 (fp16) or 6 (bf16) 16-bit products; fp32 MFMA 157.3 TFLOP/s in exact mode).  Against the fp32 MFMA "
                                 "peak the same number is frac_of_fp32_mfma_peak (> 1: the kernel removes 52% of the reference's work by "
                                 "linearity, DESIGN.md 3.3, re-uses the even coarse samples in the fine pass, 3.7, and runs the rest on the "
                                 "16-bit pipe)",
                         "field_evaluations_per_ray": {"reference": Q_PER_RAY, "executed": q_exec},
                         "mfma_executed_TFLOPs": round(EXEC_FLOP_PER_TILE[MODE] * (H * W * q_exec // 32) / (kern_ms * 1e-3) / 1e12, 2),
                         "mfma_executed_frac_of_peak": round(EXEC_FLOP_PER_TILE[MODE] * (H * W * q_exec // 32) / (kern_ms * 1e-3) /
                                                             (PEAK_FP32_MFMA if F32 else PEAK_BF16_MFMA), 4),
                         "hbm_algorithmic_bytes_per_launch": BYTES_PER_FRAME,
                         "hbm_achieved_GBps": round(BYTES_PER_FRAME / (kern_ms * 1e-3) / 1e9, 2), "hbm_frac_of_8TBps": round(BYTES_PER_FRAME / (kern_ms * 1e-3) / 8e12, 5)},
        }
        if not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            rows = args.cpu_rows or 4
            est, took = cpu_baseline(sc, rows, threads)
            if not args.cpu_rows and took < 5.0:             # scale the sample to ~15 s of CPU work
                rows = int(min(H, max(4, rows * 15.0 / max(took, 1e-3)))) // 4 * 4
                est, took = cpu_baseline(sc, rows, threads)
            res["cpu_baseline"] = {"value": round(1.0 / est, 5), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": "%d of %d image rows (%d rays) of the same frame, oracle/hav_oracle.c with OpenMP, "
                                             "%.1f s measured, scaled to a full frame" % (rows, H, rows * W, took)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
