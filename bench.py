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
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from havatar_amd import synth
    from havatar_amd.render import RayMarcher

    sc = synth.scene(8, 8, "primary")                       # constants; the rays below are the full 512x512 frame
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
    rm.set_mlp(*[t(sc["mlp"][k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
    planes = t(sc["planes"])
    vol = t(sc["vol"])
    rays = t(synth.camera_rays(H, W))[None]
    bg = torch.ones(1, H * W, 3, device=dev)
    # frame k of the batch has its own head pose; rank r renders frames r, r+N, ...
    poses = [t(synth.frame_pose((rank + world * i) % 64))[None] for i in range(8)]
    perturb = bool(args.perturb)

    def step(i):
        rm.set_triplane(planes)                              # per-frame NCHW -> channels-last re-layout (planes change per frame)
        return rm.render(rays, bg, poses[i % len(poses)], vol, S_C, S_F, perturb=perturb)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        rm.set_triplane(planes)
        ev[i][0].record()                                    # kernel time of the march on ITS stream (torch current stream)
        out = rm.render(rays, bg, poses[i % len(poses)], vol, S_C, S_F, perturb=perturb)
        ev[i][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    fps = world * args.steps / dt

    if rank == 0:
        res = {
            "metric": "rendered frames/sec @512^2, 64 samples/ray", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: one 512x512 frame per GPU per step = 262144 rays x (64 coarse + 48 fine) = 29.36M "
                                   "radiance-MLP queries; ray sampling + skinning lookup + tri-plane gather + PE + MLP + compositing + "
                                   "resampling in one launch (P5-P12); tri-plane/skin volume/MLP resident in HBM, tri-plane "
                                   "re-laid out per frame inside the step; tri-plane encoders (P3, MIOpen convs) not in the step",
                       "rays_per_frame": H * W, "num_coarse": S_C, "num_fine": S_F, "perturb": perturb,
                       "parallelism": "frames sharded, %d rank(s), no data-path collective" % world,
                       "kernel": rm.variant(S_C, S_F, perturb=perturb)},
            "roofline": {"bound": "mfma", "achieved": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA / 1e12,
                         "unit": "TFLOP/s", "frac": round(FLOP_PER_FRAME / (kern_ms * 1e-3) / PEAK_FP32_MFMA, 4), "traffic": None,
                         "kernel_ms": round(kern_ms, 3), "flop_per_launch": FLOP_PER_FRAME,
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
