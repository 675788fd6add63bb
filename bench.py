#!/usr/bin/env python3
"""bench.py -- rendered frames/s of the HAvatar ray-march hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (backend nccl == RCCL).  One "step" = one pass of the hot path over one
frame of synthetic input per rank: 512x512 rays x (64 coarse + 48 fine) MLP queries (BASELINE.json configs[1]);
frames are independent, so N ranks render N frames per step with no data-path collective ("weak" scaling).
Rank 0 prints ONE JSON line (the last line of stdout).

What the default run (N = 1) times, each as its own loop of W warm-up + K timed steps between synchronisations:
  * cfg2 in every arithmetic mode of the radiance MLP -- bf16x3 (every operand hi + mid + lo bf16 = 24 bits, six products), exact fp32
    MFMA, and the 22-bit fp16 double split.  `value` is the FASTEST MODE THAT IS NOT NARROWER THAN THE REFERENCE'S fp32; the fp16
    double split is reported under modes.fp16x2 and is never the headline.
  * extra.cfg4 (BASELINE configs[3]: the frame + SWGAN_unet 512 -> 1024), extra.cfg5 (configs[4]: train_avatar.py's step), extra.cfg3
    (configs[2] on one GPU: B = 1 / 2 / 4 frames per hipGraph replay).
  * cpu_baseline: the whole frame on the host cores -- the oracle's ray march (OpenMP) + this repo's PyTorch-CPU statement of the
    tri-plane encoders.
  * roofline.power: socket power and shader clock sampled during a sustained replay of the headline loop (the kernel runs at the
    socket power cap, docs/history/DESIGN_r1-r4.md 3.13).

Other workloads (not what the driver runs; same JSON contract):
  --workload cfg3 [--frames 64]   BASELINE configs[2]: a batch of 64 frames dealt round-robin to the ranks (8 per GPU on 8 GPUs), finished
                                  RGB frames all-gathered round by round over RCCL WHILE the next frame renders
                                  (frames.OverlappedFrameGather); one step = one batch, total work fixed -> "scaling": "strong".
  --workload cfg4                 the stage-two HD path as the headline line.
  --workload cfg5                 train_avatar.py's optimisation step as the headline line.
  --device cpu [--size 16]        plumbing mode for the tests: CPU tensors, gloo, no HIP library; exercises the N > 1 branch without GPUs.
"""
import argparse
import json
import os
import sys
import threading
import time

# ROCm hipGraph replay fault (havatar_amd/__init__.py): the runtime reads this variable when it initialises, i.e. at the process's first HIP
# call -- which in this script is torch.cuda.set_device / init_process_group, BEFORE the package is imported.  So it is set here, before
# torch is even imported; havatar_amd.hipgraph_state() reports whether the setting was in force when HIP came up.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 512                                            # (--size overrides, plumbing mode only)
S_C, S_F = 64, 16
Q_PER_RAY = S_C + (S_C + 1) // 2 + S_F                 # 112 MLP queries per ray
FLOP_PER_QUERY = 2 * (176 * 128 + 128 * 128 + 128 * 1 + 128 * 64 + 64 * 3)      # 94 848
FLOP_PER_FRAME = FLOP_PER_QUERY * Q_PER_RAY * H * W     # 2.7848e12
BYTES_PER_FRAME = 600 * H * W + 8388608 + 2097152 + 190992   # compulsory HBM bytes
PEAK_FP32_MFMA = 157.3e12                               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA = 2500e12                                # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense
DTYPE = {"fp16x2": "f32 emulated as 2 x fp16 split-operand MFMA with fp32 accumulate: 22-bit operands (hi + lo fp16, lo.lo dropped), fp16 "
                   "exponent range guarded on the device -- NARROWER than fp32: never the headline, reported as modes.fp16x2",
         "bf16x3": "f32 emulated exactly-split: every operand = hi + mid + lo bf16 = 24 significant bits (the fp32 value itself), six partial "
                   "products on v_mfma_f32_32x32x16_bf16 with fp32 accumulate (dropped terms <= 2^-23 relative; measured per-product error <= 2^-21.7): not narrower than fp32",
         "fp16x2+mx": "f32 emulated, full-width operands: every operand = hi + lo fp16 + tail (= the fp32 value itself, >= 24 significant bits); the three "
                      "leading partial products on v_mfma_f32_32x32x16_f16, the three terms of order 2^-22 (hi.tail, tail.hi, lo.lo) on block-scaled "
                      "4-/6-bit v_mfma_scale_f32_32x32x64_f8f6f4 (factors to 2-4 bits: <= 2^-25 of the product), fp32 accumulate; measured per-product error "
                      "against fp64 <= 2^-21.3 at worst with the same rms as the bf16 triple split (worst 2^-21.7; "
                      "tests/test_render_gpu.py::test_arithmetic_modes_of_the_dense_layers_against_fp64_products asserts exactly these bounds): not narrower than fp32",
         "f32": "f32 (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain)"}
# arithmetic modes of the radiance MLP (include/havatar.h): key -> (HAVATAR_MLP value, operand bits, not narrower than the reference's fp32?)
MODES = {"fp16x2+mx": ("mx", 24, True), "bf16x3": ("split", 24, True), "f32": ("f32", 24, True), "fp16x2": ("half", 22, False)}
DEFAULT_MODE = "fp16x2+mx"          # what the Python layer runs when HAVATAR_MLP is unset (havatar_amd/render.py::DEFAULT_MLP)
MODE_OF_ENV = {"mx": "fp16x2+mx", "split": "bf16x3", "bf16": "bf16x3", "half": "fp16x2", "f32": "f32"}
# matrix-core work the kernel executes per 32-sample tile: 11 k-chunks x 4 row tiles x 3 (fp16) / 6 (bf16) products of a
# 16-bit 32x32x16 MFMA (32768 FLOP each), or 352 v_mfma_f32_32x32x2_f32 (4096 FLOP each) in the exact-fp32 mode
# (fp16x2+mx: 132 fp16 products + 36 block-scaled 32x32x64 instructions of 131072 FLOP each, priced against the 16-bit peak like the rest)
EXEC_FLOP_PER_TILE = {"fp16x2": 132 * 32768, "bf16x3": 264 * 32768, "f32": 352 * 4096, "fp16x2+mx": 132 * 32768 + 36 * 131072}
EXEC_INSTR_PER_TILE = {"fp16x2": 132, "bf16x3": 264, "f32": 352, "fp16x2+mx": 168}
# feature parking (fp16 cache kernels, docs/history/DESIGN_r1-r4.md 3.7): the 48 parked tiles of the 80 evaluated per ray block also run fc_rgbFeat on
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


class PowerSampler:
    """Socket power (W) and shader clock (GHz) while a loop runs: amdgpu's hwmon / sysfs files, read every few ms from a thread."""

    def __init__(self, index=0):
        """index: the torch device index.  The sysfs node is found through the device's PCI address (a box shows every GPU of the node in
        sysfs, also the ones this process cannot see); failing that, every amdgpu node is sampled and the busiest one is reported."""
        import glob
        self.nodes = []                       # [(power file, clock file, cap file)]
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cand = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                continue
            bdf = os.path.basename(os.path.realpath(dev))
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                pf = next((os.path.join(hw, nm) for nm in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, nm))), None)
                cf = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
                kf = os.path.join(hw, "power1_cap") if os.path.exists(os.path.join(hw, "power1_cap")) else None
                if pf or cf:
                    cand.append((bdf, pf, cf, kf))
        exact = [c for c in cand if want and c[0].lower() == want.lower()]
        self.matched = bool(exact)
        self.nodes = [c[1:] for c in (exact or cand)]
        self.samples, self._stop, self._th = [[] for _ in self.nodes], False, None

    @staticmethod
    def _read(f):
        try:
            return float(open(f).read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def __enter__(self):
        def run():
            while not self._stop:
                for i, (pf, cf, _) in enumerate(self.nodes):
                    pw = self._read(pf) if pf else None
                    ck = self._read(cf) if cf else None
                    if pw is not None or ck is not None:
                        self.samples[i].append((pw, ck))
                time.sleep(0.004)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._th.join(timeout=1.0)

    def summary(self):
        best = None
        for (pf, cf, kf), smp in zip(self.nodes, self.samples):
            pw = [p * 1e-6 for p, _ in smp if p]
            ck = [c * 1e-9 for _, c in smp if c]
            if not pw and not ck:
                continue
            pw, ck = pw[len(pw) // 3:], ck[len(ck) // 3:]          # the first third of the window is the ramp
            mean_pw = sum(pw) / len(pw) if pw else 0.0
            if best is None or mean_pw > best[0]:
                best = (mean_pw, pw, ck, kf, len(smp))
        if best is None:
            return None
        _, pw, ck, kf, n = best
        cap = self._read(kf) if kf else None
        return {"socket_power_W": round(sum(pw) / len(pw), 1) if pw else None, "socket_power_max_W": round(max(pw), 1) if pw else None,
                "power_cap_W": round(cap * 1e-6, 1) if cap else None, "sclk_GHz": round(sum(ck) / len(ck), 3) if ck else None,
                "sclk_max_GHz": 2.4, "samples": n,
                "source": "amdgpu hwmon (power1_average, freq1_input) during a sustained replay; node %s" % (
                    "matched by PCI address" if self.matched else "= the busiest of the %d amdgpu nodes in sysfs" % len(self.nodes))}


def live_pmc_traffic(mlp_env):
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
    env = dict(os.environ, TMPDIR="/tmp", HAVATAR_MLP=mlp_env)
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


def cpu_march(sc, rows, threads):
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


def cpu_encoders(cfg, threads):
    """P3 on the host cores: this repo's PyTorch statement of set_conditional_embedding (two StyleGAN_zxc encoders, 327 GFLOP of fp32
    convolutions; reference model/nerf_model.py:58-86) on CPU tensors -- the same module code the CPU parity tests pin to the reference.
    oneDNN's convolutions do not scale to every hardware thread of a 2-socket host (256 threads: 45 s per call, measured), so the call is
    timed at a few thread counts and the best one is reported with its count."""
    import numpy as np
    import torch
    from havatar_amd import synth
    from havatar_amd.model.nerf_trainer import Trainer
    old = torch.get_num_threads()
    try:
        torch.manual_seed(0)
        tr = Trainer(cfg, 1)
        tr.requires_grad_(False)
        synth.fill_state_dict(tr)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
        front, left, right = [t(a) for a in synth.cond_images()]
        pose = t(synth.frame_pose(0))[None]
        best = None
        tried = {}
        for nt in sorted({n for n in (16, 32, 64, threads // 2) if 1 <= n <= threads}):
            torch.set_num_threads(nt)
            ts = []
            with torch.no_grad():
                for _ in range(3):                       # first call pays oneDNN primitive creation
                    t0 = time.perf_counter()
                    tr.model_coarse.set_conditional_embedding(front_render_cond=front, left_render_cond=left, right_render_cond=right,
                                                              latents=tr.latent_codes[0:1], cond_c=pose.view(1, -1))
                    ts.append(time.perf_counter() - t0)
                    if ts[-1] > 12.0:
                        break
            tried[nt] = round(min(ts[1:] or ts), 3)
            if best is None or tried[nt] < best[0]:
                best = (tried[nt], nt)
        return best[0], best[1], tried
    finally:
        torch.set_num_threads(old)


def run_cfg5(args, emit=True):
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
    if emit:
        emit_line(res)
    return res


def make_upsampler(args, render, poses, dev):
    """Stage two on top of the frame (reference: avatarHD_reenactment.py:153-160): SWGAN_unet(styles=[style], condition_img=render[:, 3:])."""
    import torch
    from havatar_amd import synth
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
        return lambda feat: up_g(condition_img=feat)

    def upsample(feat):
        with torch.no_grad():
            return up(styles=[style], condition_img=feat)
    return upsample


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="image rows timed on the CPU (0 = auto: the whole frame if it fits ~25 s)")
    ap.add_argument("--perturb", type=int, default=1, help="stratified jitter on (reference default for inference)")
    ap.add_argument("--graph", type=int, default=1, help="replay the frame as one hipGraph (0 = eager launches)")
    ap.add_argument("--live-pmc", type=int, default=1, help="measure roofline.traffic in this run (2 rocprofv3 --pmc passes, ~1 min; N=1 only)")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2")
    ap.add_argument("--extras", type=int, default=1, help="N = 1, cfg2: also time every arithmetic mode, cfg4 and cfg5 (extra.*) in this run")
    ap.add_argument("--frames", type=int, default=64, help="cfg3: frames per batch (one step = one batch)")
    ap.add_argument("--frame-batch", type=int, default=0,
                    help="cfg3: frames per hipGraph replay (the encoders see a batch, the march B x R rays in one launch).  0 = at N = 1 time "
                         "B = 1, 2 and 4 and report the fastest (config.frames_per_replay holds all three); at N > 1: 4")
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
        args.graph, args.live_pmc, args.no_cpu_baseline, args.extras = 0, 0, True, 0
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
    # what the process group itself reports, for the driver's SCALE record: every rank contributes (1, its device index) to one all-reduce
    ranks_seen = None
    if dist.is_initialized():
        probe = torch.tensor([1.0, float(local)], dtype=torch.float64, device=dev if not cpu else "cpu")
        dist.all_reduce(probe)
        ranks_seen = {"ranks": int(probe[0].item()), "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "sum_of_local_device_indices": int(probe[1].item()),
                      "devices_visible_to_this_rank": 0 if cpu else torch.cuda.device_count(),
                      "visible_devices_env": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}
    sync = (lambda: None) if cpu else torch.cuda.synchronize

    from havatar_amd import hipgraph_state, synth
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

    def make_render(fb=1):
        """The frame as a callable pose -> (render, mask, ...) in the marcher's CURRENT arithmetic mode (re-captured per mode).
        fb > 1: `fb` frames per call -- poses [fb,4,3] -> render [fb,67,H,W] (every frame of the batch with its own pose; the synthetic
        condition images are the same for all frames, as in the one-frame loop)."""
        d = data if fb == 1 else {k: (v.expand(fb, *v.shape[1:]).contiguous() if torch.is_tensor(v) else v) for k, v in data.items()}
        if args.graph:
            from havatar_amd.graph import GraphedForward
            frame = GraphedForward(tr, d)                      # the whole frame = one hipGraph launch (+ the pose copy)
            return lambda pose: frame(inv_head_T=pose)

        def eager(pose):
            with torch.no_grad():
                return tr(**{**d, "inv_head_T": pose})
        return eager

    def timed_loop(step, frames_per_step):
        """W untimed steps, then EXACTLY K timed steps between barrier + synchronize on both sides; MAX over ranks."""
        out = None
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
        return dt, frames_per_step * args.steps / dt, out

    # ---- which arithmetic mode(s) ------------------------------------------------------------------------------------
    env_mode = MODE_OF_ENV.get(os.environ.get("HAVATAR_MLP", ""), None)
    all_modes = (not cpu) and world == 1 and args.workload == "cfg2" and bool(args.extras) and env_mode is None
    if cpu:
        mode_list = ["f32"]
    elif all_modes:
        mode_list = list(MODES)                                 # bf16x3, f32, fp16x2: each gets its own timed loop
    else:
        mode_list = [env_mode or DEFAULT_MODE]
    m = None
    if not cpu:
        from havatar_amd import _lib
        MLP_CONST = {"bf16x3": _lib.HAV_MLP_SPLIT_BF16, "f32": _lib.HAV_MLP_F32, "fp16x2": _lib.HAV_MLP_SPLIT_F16, "fp16x2+mx": _lib.HAV_MLP_SPLIT_F16_MX}
        m = tr._hip_marcher()

    def set_mode(mode):
        if m is not None:
            m.mlp_mode = MLP_CONST[mode]

    upsample = None
    gather = None
    out = None
    loops = {}
    frames_per_replay = None
    for mode in mode_list:
        set_mode(mode)
        render = make_render()
        if args.workload == "cfg3":
            # one step = one batch of --frames frames: this rank renders frames rank, rank + N, ...; the finished RGB frame of round r is
            # all-gathered (RCCL over xGMI) while round r + 1 renders; every rank ends the step holding the whole batch.  Throughput
            # mode: FB of this rank's frames per hipGraph replay (avatarHD_reenactment.py:149-170 is a loop over independent frames and
            # model/nerf_model.py:58-86 takes a batch), each still handed to the exchange as its own round.
            from havatar_amd.frames import OverlappedFrameGather
            gather = OverlappedFrameGather(args.frames, (3, H, W), device=dev, force_collective=bool(args.force_collective))
            batch_poses = {k: t(synth.frame_pose(k % 64))[None] for k in range(rank, args.frames, world)}

            def make_step(fb, render1=render):
                render_fb = make_render(fb) if fb > 1 else None

                def step(i):
                    r = 0
                    while r < gather.rounds:
                        ks = [gather.my_frame(r + b) for b in range(fb) if r + b < gather.rounds]
                        if fb > 1 and len(ks) == fb and None not in ks:
                            out = render_fb(torch.cat([batch_poses[k] for k in ks]))[0]
                            for b in range(fb):
                                gather.submit(r + b, out[b, :3])
                            r += fb
                        else:                                   # ragged tail of the batch: one frame per replay
                            k = gather.my_frame(r)
                            gather.submit(r, None if k is None else render1(batch_poses[k])[0][0, :3])
                            r += 1
                    return gather.finalize()
                return step
            fbs = [args.frame_batch] if args.frame_batch > 0 else ([1, 2, 4] if (world == 1 and not cpu) else [1 if cpu else 4])
            frames_per_step = args.frames
            frames_per_replay = {}
            best = None
            for fb in fbs:
                st = make_step(fb)
                dt_, fps_, out_ = timed_loop(st, frames_per_step)
                frames_per_replay[str(fb)] = {"frames_per_s": round(fps_, 3), "ms_per_frame": round(1e3 * dt_ / (args.steps * args.frames) * world, 3)}
                if best is None or fps_ > best[1]:
                    best = (dt_, fps_, out_, st, fb)
            dt, fps, out, step = best[:4]
            frames_per_replay["used"] = best[4]
            loops[mode] = {"dt": dt, "fps": fps, "render": render, "step": step, "frames_per_step": frames_per_step}
            continue
        elif args.workload == "cfg4":
            upsample = make_upsampler(args, render, poses, dev)

            def step(i, render=render):
                return upsample(render(poses[i % len(poses)])[0][:, 3:])
            frames_per_step = world
        else:
            def step(i, render=render):
                return render(poses[i % len(poses)])
            frames_per_step = world
        dt, fps, out = timed_loop(step, frames_per_step)
        loops[mode] = {"dt": dt, "fps": fps, "render": render, "step": step, "frames_per_step": frames_per_step}

    if cpu:
        # plumbing mode: no kernels to profile; report the contract fields and what the collective moved
        dt, fps = loops["f32"]["dt"], loops["f32"]["fps"]
        if rank == 0:
            emit_line(({"metric": "rendered frames/sec @%d^2, 64 samples/ray" % H, "value": round(fps, 4), "unit": "frames/s",
                              "n_gpus": 0, "ranks": world, "rccl_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
                              "scaling": "strong" if args.workload == "cfg3" else "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": args.workload + " (CPU plumbing mode: PyTorch statement of the path, gloo)",
                                                               "frames_per_step": loops["f32"]["frames_per_step"], "size": H,
                                                               "gathered": list(out.shape) if args.workload == "cfg3" else None}}))
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- the headline: the fastest timed mode that is NOT narrower than the reference's fp32 --------------------------------
    wide = [k for k in loops if MODES[k][2]]
    head = max(wide, key=lambda k: loops[k]["fps"]) if wide else mode_list[0]
    dt, fps, frames_per_step = loops[head]["dt"], loops[head]["fps"], loops[head]["frames_per_step"]

    # ---- outside the timed region: per-phase device times (HIP events on the launch stream = torch's current stream) ----
    def timed(fn, n):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a_, b_ in ev:
            a_.record(); fn(); b_.record()
        torch.cuda.synchronize()
        return float(np.median([a_.elapsed_time(b_) for a_, b_ in ev]))

    vol = tr.headpose_skin_net.current_volume()
    n_ev = max(3, min(args.steps, 10))
    kern_ms, variants = {}, {}
    with torch.no_grad():
        # (the last timed replay may have been a batched one: the planes of ONE frame for the per-phase timings below)
        tr.model_coarse.set_conditional_embedding(front_render_cond=front, left_render_cond=left, right_render_cond=right,
                                                  latents=tr.latent_codes[0:1], cond_c=poses[0].view(1, -1))
        for mode in mode_list:
            set_mode(mode)
            m.set_mlp(*[t_.detach() for t_ in tr.model_coarse.mlp_tensors()])
            m.set_triplane(tr.model_coarse.triPlane_embeddings.detach())
            kern_ms[mode] = timed(lambda: m.render(rays, bg, poses[0], vol, S_C, S_F, perturb=perturb, coarse_outputs=False), n_ev if mode == head else 5)
            variants[mode] = m.last_variant
        set_mode(head)
        prep_ms = timed(lambda: m.set_triplane(tr.model_coarse.triPlane_embeddings), n_ev)
        enc_ms = timed(lambda: tr.model_coarse.set_conditional_embedding(
            front_render_cond=front, left_render_cond=left, right_render_cond=right, latents=tr.latent_codes[0:1],
            cond_c=poses[0].view(1, -1)), n_ev)

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
        if "SQ_WAIT_ANY" in k and "SQ_WAVE_CYCLES" in k:
            u["waves_waiting"] = round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 3)
        u["source"] = src
        return u

    def march_roofline(mode, kms):
        """Roofline block of the march kernel in one arithmetic mode (kms = its launch time by HIP events)."""
        kname = variants[mode]
        cached = kname.endswith((", 1>", ", 2>"))              # fine-pass cache: the 32 even coarse samples are not evaluated again
        q_exec = (S_C + S_F) if cached else Q_PER_RAY
        tiles = H * W * q_exec // 32
        exec_flop = EXEC_FLOP_PER_TILE[mode] * tiles
        if mode == "fp16x2" and cached:                        # feature parking: fc_rgbFeat on the matrix cores for the 48 parked tiles of a block
            exec_flop += PARK_FLOP_PER_TILE * (H * W // 32) * ((S_C + 1) // 2 + S_F)
        peak = PEAK_FP32_MFMA if mode == "f32" else PEAK_BF16_MFMA
        return {"kernel": kname, "kernel_ms": round(kms, 3), "achieved": round(FLOP_PER_FRAME / (kms * 1e-3) / 1e12, 3), "peak": peak / 1e12,
                "frac": round(FLOP_PER_FRAME / (kms * 1e-3) / peak, 4), "field_evaluations_per_ray": {"reference": Q_PER_RAY, "executed": q_exec},
                "mfma_executed_TFLOPs": round(exec_flop / (kms * 1e-3) / 1e12, 2), "mfma_executed_frac_of_peak": round(exec_flop / (kms * 1e-3) / peak, 4),
                "mfma_instructions_per_launch_model": EXEC_INSTR_PER_TILE[mode] * tiles + (48 * (H * W // 32) * ((S_C + 1) // 2 + S_F) if (mode == "fp16x2" and cached) else 0),
                "tiles": tiles}

    if rank == 0:
        kname = variants[head]
        kms = kern_ms[head]
        traffic = (live_pmc_traffic(MODES[head][0]) if (args.live_pmc and world == 1) else None) or pmc_traffic(kname)
        busy = unit_busy(kname)
        rf = march_roofline(head, kms)
        # which unit is busiest (committed counters of this variant): the label the fraction below must be read with
        names = {"ta": "ta (texture addresser = the L1 gather path of the 8 tri-plane taps)", "mfma": "mfma", "valu": "valu"}
        busiest = max((k for k in ("ta", "mfma", "valu") if busy and k in busy), key=lambda k: busy[k], default=None)
        # the tap path gets its own roofline: bytes the 8 tri-plane taps move into VGPRs per launch (128 x 16 B per lane and evaluated
        # tile: the byte count is fixed by the algebra of DESIGN.md 3.3) against the texture path's 64 B/clk/CU
        tap_bytes = 128 * 16 * 64 * rf["tiles"]
        # sustained replay of the headline loop with the socket power / shader clock sampled (the timed region above is K steps only)
        power = None
        if world == 1 and args.workload == "cfg2":
            with PowerSampler(local) as ps:
                t0 = time.perf_counter()
                n_sus = 0
                while time.perf_counter() - t0 < 2.5:
                    for i in range(25):
                        loops[head]["step"](i)
                    torch.cuda.synchronize()
                    n_sus += 25
                sus_dt = time.perf_counter() - t0
            power = ps.summary()
            if power is not None:
                power.update(sustained_steps=n_sus, sustained_ms_per_step=round(1e3 * sus_dt / n_sus, 3))
        clk_used = (power or {}).get("sclk_GHz") or 2.4
        ta_peak = 64.0 * 256 * clk_used * 1e9
        ta_roof = {"what": "tri-plane tap bytes delivered to VGPRs per launch / (64 B/clk/CU x 256 CUs x clock)", "bytes_per_launch": tap_bytes,
                   "achieved_TBps": round(tap_bytes / (kms * 1e-3) / 1e12, 2), "peak_TBps": round(ta_peak / 1e12, 2), "clock_GHz": round(clk_used, 3),
                   "clock_source": "sclk sampled during the sustained replay" if (power or {}).get("sclk_GHz") else "2.4 GHz maximum clock",
                   "frac": round(tap_bytes / (kms * 1e-3) / ta_peak, 4), "busy": (busy or {}).get("ta")}
        cfg4_ms = None
        if args.workload == "cfg4":          # the upsampler alone (its own graph), HIP events around the replay
            with torch.no_grad():
                feat = loops[head]["render"](poses[0])[0][:, 3:].contiguous()
                cfg4_ms = timed(lambda: upsample(feat), n_ev)
        step_ms = 1e3 * dt / args.steps / (frames_per_step / world)               # per frame of this rank
        power_bound = bool(power and power.get("sclk_GHz") and power["sclk_GHz"] < 2.3)
        res = {
            "metric": ("stage-two HD frames/sec: 512^2 NeRF volume render (64 samples/ray) + SWGAN_unet upsampler to 1024^2" if args.workload == "cfg4"
                       else "rendered frames/sec @512^2, 64 samples/ray"), "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "rccl_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if args.workload == "cfg3" else "weak", "vs_baseline": None, "dtype": DTYPE[head],
            "data": "synthetic",
            "config": {"batch": CFG3_NOTE % (args.frames, world, -(-args.frames // world)) if args.workload == "cfg3" else None,
                       "workload": "cfg2: Trainer.forward(render_full_img=True) for one 512x512 frame per GPU per step: tri-plane encoders "
                                   "(P3: 2x StyleGAN_zxc: 3x3 / up-sampling convolutions on split-fp16 MFMA HIP kernels with the block glue fused, stride-2 and 1x1 ones on MIOpen, HIP upfirdn2d/fused_bias_act) -> per-frame plane projection -> fused "
                                   "ray march (P5-P12) over 262144 rays x (64 coarse + 48 fine) = 29.36M radiance-MLP queries -> [1,67,512,512]",
                       "arithmetic_mode": head, "operand_bits": MODES[head][1],
                       "headline_rule": "value = the fastest TIMED mode whose operands are not narrower than the reference's fp32 (modes.* holds every timed loop)",
                       "phase_ms": {"encoders_P3": round(enc_ms, 3), "plane_prepare": round(prep_ms, 3), "ray_march_kernel": round(kms, 3),
                                    "encoders_P3_inside_the_graph": round(step_ms - kms - prep_ms, 3),
                                    "note": "encoders_P3_inside_the_graph is a DIFFERENCE (step - march - preparation) and inherits the march's box-to-box and "
                                            "clock-state spread: the march is timed alone here, straight after an idle period, and runs 1-2 % slower at the "
                                            "sustained power cap inside the frame loop (profiles/r05_frame_streams.txt holds the per-queue trace of a replay: "
                                            "the encoder phase ends 1.72-1.82 ms after the previous march).  encoders_P3 is timed eagerly on one stream; inside the frame's hipGraph the two encoders run "
                                            "on two streams without launch gaps: step - march - preparation"},
                       "stage_two": ({"upsampler_ms": round(cfg4_ms, 3), "output": [1, 3, 2 * H, 2 * W],
                                      "what": "BASELINE configs[3]: the step is the cfg2 frame below followed by SWGAN_unet(512 -> 1024) on render[:, 3:] "
                                              "(avatarHD_reenactment.py:153-160): its 3x3 / up-sampling convolutions, Haar transforms, upfirdn2d and "
                                              "fused_bias_act are kernels of this library, stride-2 / 1x1 convolutions MIOpen; phase_ms.encoders_P3_inside_the_graph "
                                              "includes the upsampler in this workload"} if args.workload == "cfg4" else None),
                       "rays_per_frame": H * W, "num_coarse": S_C, "num_fine": S_F, "perturb": perturb, "hipgraph": bool(args.graph), "frames_per_replay": frames_per_replay,
                       "hipgraph_packet_capture": hipgraph_state(),          # in_force = the variable was 0 before this process's first HIP call (ROCm graph-replay fault, havatar_amd/__init__.py)
                       "parallelism": ("frames sharded, %d rank(s), one overlapped all_gather of finished frames per round" if args.workload == "cfg3"
                                       else "frames sharded, %d rank(s), no data-path collective") % world,
                       "exchange": (("RCCL all_gather_into_tensor from a side stream, %d-rank group%s" % (world, " (forced at N = 1)" if world == 1 else ""))
                                    if (args.workload == "cfg3" and gather is not None and gather.collective) else None),
                       "kernel": kname},
            "roofline": {"bound": "mfma",
                         "limited_by": ("package power (PPT): under this kernel's load the firmware holds the shader clock below its 2.4 GHz maximum "
                                        "(roofline.power; profiles/r04_throttle_status.txt: PPT violation active, thermal limiters idle; DESIGN.md "
                                        "3.13: cycle savings come back as lower clock)" if power_bound else busiest),
                         "busiest_unit": names.get(busiest), "unit_busy": busy, "ta": ta_roof, "power": power,
                         "achieved": rf["achieved"], "peak": rf["peak"], "unit": "TFLOP/s", "frac": rf["frac"],
                         "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic,
                         "kernel_ms": rf["kernel_ms"], "flop_per_launch": FLOP_PER_FRAME,
                         "algebraic_reduction": {"algorithmic_flop_rate_over_fp32_mfma_peak": round(FLOP_PER_FRAME / (kms * 1e-3) / PEAK_FP32_MFMA, 4),
                                                 "what": "NOT a fraction of a roofline: the reference network's FLOP per second over the fp32 matrix peak.  It exceeds 1 because "
                                                         "the kernel does not execute those FLOP: layer 1's 128 plane columns are folded into the planes once per frame, the "
                                                         "feature head is applied to composited hidden units, the fine pass re-uses the even coarse samples (80 of 112 "
                                                         "evaluations), and what is left runs on the 16-bit pipe; the fraction of a peak is mfma_executed_frac_of_peak"},
                         "note": "the path's only dense contraction is the radiance MLP, so the roofline is priced in FLOP ('bound': mfma): "
                                 "achieved = ALGORITHMIC fp32 FLOP of the reference network (94848/query x 112 x 262144) / kernel time; peak = the "
                                 "dense peak of the matrix pipe the kernel runs on (16-bit MFMA 2.5 PFLOP/s in the split modes, where one fp32 "
                                 "product costs 3 (fp16) or 6 (bf16) 16-bit products; fp32 MFMA 157.3 TFLOP/s in exact mode).  The matrix cores "
                                 "are NOT what limits the kernel: it draws the socket's power cap and runs below the maximum clock (power), "
                                 "and among the units the texture addresser serving the tri-plane gather is the busiest (unit_busy, ta); "
                                 "mfma_executed_* counts what the matrix cores really execute",
                         "field_evaluations_per_ray": rf["field_evaluations_per_ray"],
                         "mfma_executed_TFLOPs": rf["mfma_executed_TFLOPs"], "mfma_executed_frac_of_peak": rf["mfma_executed_frac_of_peak"],
                         "mfma_instructions_per_launch_model": rf["mfma_instructions_per_launch_model"],
                         "hbm_algorithmic_bytes_per_launch": BYTES_PER_FRAME,
                         "hbm_achieved_GBps": round(BYTES_PER_FRAME / (kms * 1e-3) / 1e9, 2), "hbm_frac_of_8TBps": round(BYTES_PER_FRAME / (kms * 1e-3) / 8e12, 5)},
        }
        # every arithmetic mode that was timed: its own loop (same W / K), its own kernel time
        res["modes"] = {}
        for mode in mode_list:
            r_ = march_roofline(mode, kern_ms[mode])
            res["modes"][mode] = {"frames_per_s": round(loops[mode]["fps"], 3), "ms_per_step": round(1e3 * loops[mode]["dt"] / args.steps, 3),
                                  "timed_loop": True, "operand_bits": MODES[mode][1], "not_narrower_than_fp32": MODES[mode][2],
                                  "kernel": r_["kernel"], "kernel_ms": r_["kernel_ms"],
                                  "algorithmic_flop_rate_over_peak": r_["frac"],          # (> 1 in the exact-f32 mode: an algebraic-reduction factor, roofline.algebraic_reduction)
                                  "mfma_executed_frac_of_peak": r_["mfma_executed_frac_of_peak"], "dtype": DTYPE[mode]}
        res["extra"] = {}
        if world == 1 and args.workload == "cfg2" and args.extras:
            # BASELINE configs[3] and [4] under the same clock as the headline: their own loops, same --steps / --warmup
            set_mode(head)
            try:
                rend = loops[head]["render"]
                up = make_upsampler(args, rend, poses, dev)
                dt4, fps4, _ = timed_loop(lambda i: up(rend(poses[i % len(poses)])[0][:, 3:]), 1)
                with torch.no_grad():
                    feat = rend(poses[0])[0][:, 3:].contiguous()
                    up_ms = timed(lambda: up(feat), n_ev)
                res["extra"]["cfg4"] = {"frames_per_s": round(fps4, 3), "ms_per_step": round(1e3 * dt4 / args.steps, 3), "upsampler_ms": round(up_ms, 3),
                                        "arithmetic_mode": head, "output": [1, 3, 2 * H, 2 * W],
                                        "roofline": {"kernel": "SWGAN_unet(512 -> 1024) as one hipGraph (352 GFLOP of fp32 convolutions: 3x3 and up-sampling ones on "
                                                               "split-fp16 MFMA HIP kernels, stride-2 / 1x1 on MIOpen; 66 upfirdn2d + 38 fused_bias_act calls)",
                                                     "bound": "mfma", "achieved": round(352e9 / (up_ms * 1e-3) / 1e12, 1), "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                                                     "frac": round(352e9 / (up_ms * 1e-3) / PEAK_BF16_MFMA, 4),
                                                     "custom_op_bytes": int(816.7e6 + 1005.1e6),
                                                     "custom_op_floor_ms_at_8TBps": round((816.7e6 + 1005.1e6) / 8e12 * 1e3, 3)},
                                        "what": "BASELINE configs[3]: the cfg2 frame + SWGAN_unet on render[:, 3:] (avatarHD_reenactment.py:153-160), both hipGraphs"}
            except Exception as e:          # an extra must never cost the headline line
                res["extra"]["cfg4"] = {"error": repr(e)[:300]}
            # BASELINE configs[2] at N = 1 (its exchange degenerates to a frame copy): a batch of 16 frames per step, B = 1 / 2 / 4 frames per
            # hipGraph replay (throughput mode: the encoders see a batch, the march B x R rays per launch); bench.py --workload cfg3 is the full loop
            try:
                set_mode(head)
                c3 = {}
                n3 = 16
                for fb in (1, 2, 4):
                    rfb = loops[head]["render"] if fb == 1 else make_render(fb)
                    pz = [torch.cat([poses[(j + b) % len(poses)] for b in range(fb)]) for j in range(0, n3, fb)]

                    def step3(i, rfb=rfb, pz=pz):
                        o = None
                        for pp in pz:
                            o = rfb(pp)
                        return o
                    dt3, fps3, _ = timed_loop(step3, n3)
                    c3[str(fb)] = {"frames_per_s": round(fps3, 3), "ms_per_frame": round(1e3 * dt3 / (args.steps * n3), 3)}
                    del rfb
                res["extra"]["cfg3"] = {"frames_per_replay": c3, "frames_per_step": n3, "arithmetic_mode": head, "n_gpus": 1,
                                        "what": "BASELINE configs[2] on ONE GPU: frames per second over a sequence of frames with B frames per hipGraph replay "
                                                "(tests/test_frames_gpu.py::test_batched_frames_equal_the_frames_rendered_one_at_a_time: march bit-exact, frame <= 1e-3)"}
            except Exception as e:
                res["extra"]["cfg3"] = {"error": repr(e)[:300]}
            try:
                r5 = run_cfg5(args, emit=False)
                res["extra"]["cfg5"] = {"steps_per_s": r5["value"], "ms_per_step": r5["ms_per_step"], "dtype": r5["dtype"], "roofline": r5["roofline"],
                                        "config": r5["config"], "what": "BASELINE configs[4]: train_avatar.py's optimisation step (train_avatar.py:106-158), one hipGraph launch"}
            except Exception as e:
                res["extra"]["cfg5"] = {"error": repr(e)[:300]}
            # P13 / P14 at the cfg4 sizes under the driver's clock: tools/bench_ops.py --brief in its own process (every timed launch on its own
            # buffers, >= 1 GiB of distinct memory per timed sequence: DRAM bandwidth, not the Infinity Cache)
            try:
                import subprocess
                r_ops = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_ops.py"), "--brief"], capture_output=True, text=True, timeout=240)
                rows_ops = json.loads(r_ops.stdout.strip().splitlines()[-1])
                res["extra"]["ops"] = {"rows": rows_ops, "bound": "hbm", "peak_GBps": 8000,
                                       "what": "fused_bias_act (P13) and upfirdn2d (P14) at BASELINE configs[3] sizes, f32: algorithmic bytes (in + out) / kernel "
                                               "time (HIP events around a hipGraph of K back-to-back launches on K distinct buffer sets)"}
            except Exception as e:
                res["extra"]["ops"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:          # (rank 0 at N = 1 only: at N > 1 the other ranks would sit in the final barrier)
            threads = os.cpu_count() or 1
            rows = args.cpu_rows or 8
            est, took = cpu_march(sc, rows, threads)
            if not args.cpu_rows and took < 8.0:             # the whole frame if it fits ~25 s, else ~15 s of rows
                rows = H if est < 25.0 else int(min(H, max(8, rows * 15.0 / max(took, 1e-3)))) // 4 * 4
                est, took = cpu_march(sc, rows, threads)
            # one core: rows scaled so that it is ~10 s of work for one thread
            rows1 = 2
            est1, took1 = cpu_march(sc, rows1, 1)
            if took1 < 5.0:
                rows1 = int(max(2, min(H, rows1 * 10.0 / max(took1, 1e-3))))
                est1, took1 = cpu_march(sc, rows1, 1)
            enc_s, enc_threads, enc_tried = cpu_encoders(cfg, threads)
            res["cpu_baseline"] = {"value": round(1.0 / (est + enc_s), 5), "unit": "frames/s", "cores": threads, "kind": "port",
                                   "scope": "the whole frame, as the GPU step: tri-plane encoders (P3) + ray march (P5-P12)",
                                   "cpu_model": cpu_model(),
                                   "sample": "march: %d of %d image rows (%d rays) of the same frame, oracle/hav_oracle.c with OpenMP on %d threads, %.1f s "
                                             "measured%s; encoders: this repo's PyTorch-CPU statement of set_conditional_embedding (the module the CPU parity "
                                             "tests pin to the reference), best of 2 calls after a warm-up at the best of the thread counts tried (%s): "
                                             "%.2f s on %d threads" % (rows, H, rows * W, threads, took, "" if rows == H else ", scaled to a full frame",
                                                                        ", ".join("%d: %.2f s" % kv for kv in sorted(enc_tried.items())), enc_s, enc_threads),
                                   "encoder_threads": enc_threads,
                                   "parts_s": {"ray_march": round(est, 3), "encoders": round(enc_s, 3)},
                                   "ray_march_only": {"value": round(1.0 / est, 5), "unit": "frames/s", "cores": threads},
                                   "one_core": {"value": round(1.0 / est1, 6), "unit": "frames/s", "cores": 1, "scope": "ray march only",
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
