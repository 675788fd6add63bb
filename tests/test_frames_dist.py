"""CPU, 2 processes, gloo: the frame-sharded path (BASELINE config 3) -- every rank renders its frames with the Trainer
mirror (PyTorch path on CPU tensors), one all_gather assembles the batch, result == the single-process batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame(k):
    """a tiny deterministic 'render' of frame k through the oracle-checked torch path: 4x4 rays, 16+8 samples"""
    sys.path.insert(0, ROOT)
    from havatar_amd import synth
    from havatar_amd.utils.nerf_util import volume_render_radiance_field
    g = torch.Generator().manual_seed(100 + k)
    rf = torch.randn(16, 12, 5, generator=g)
    z = torch.sort(torch.rand(16, 12, generator=g), -1)[0] + 3.0
    d = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1)
    rgb, _, acc, _, _ = volume_render_radiance_field(rf, z, d, background_prior=torch.ones(16, 3))
    return torch.cat([rgb, acc[:, None]], -1) + float(np.linalg.det(synth.frame_pose(k)[:3].astype(np.float64)))


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from havatar_amd.frames import render_frame_batch, shard_frames
    out = render_frame_batch(_frame, n_frames)
    mine = shard_frames(n_frames, rank, world)
    q.put((rank, out.numpy(), mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_frame_sharding_two_ranks_gloo(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.stack([_frame(k) for k in range(n_frames)], 0).numpy()
    seen = []
    for rank, out, mine in res:
        assert out.shape == ref.shape and np.array_equal(out, ref)
        seen += mine
    assert sorted(seen) == list(range(n_frames))          # every frame rendered exactly once


def _trainer_worker(rank, world, port, n_frames, q):
    """Every rank builds the SAME Trainer (key-derived weights) and renders its frames of the batch through the real CPU path
    (Trainer.forward: encoders -> planes -> march), handing each finished frame to OverlappedFrameGather."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from havatar_amd.frames import OverlappedFrameGather
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    render_one, S = _trainer_renderer()
    g = OverlappedFrameGather(n_frames, (3, S, S), device="cpu")
    for r in range(g.rounds):
        k = g.my_frame(r)
        g.submit(r, None if k is None else render_one(k))
    out = g.finalize()
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _trainer_renderer(S=6):
    from havatar_amd import synth
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config(render_size=S, gen_size=4 * S, img_res=S))
    v = cfg.nerf.validation
    v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = 16, 8, False, 0.0
    torch.manual_seed(0)
    tr = synth.fill_state_dict(Trainer(cfg, 1).requires_grad_(False)).eval()
    tr.headpose_skin_net.fix_canonical_W()
    front, left, right = [torch.from_numpy(a) for a in synth.cond_images()]
    rays = torch.from_numpy(synth.camera_rays(S, S))[None]
    bg = torch.ones(1, S * S, 3)

    def render_one(k):
        with torch.no_grad():
            render, _, _ = tr(ray_batch=rays, background_prior=bg, inv_head_T=torch.from_numpy(synth.frame_pose(k))[None],
                              front_render_cond=front, left_render_cond=left, right_render_cond=right, mode="validation", fidx=0,
                              render_full_img=True)
        return render[0, :3].contiguous()
    return render_one, S


@pytest.mark.parametrize("n_frames", [3])
def test_real_trainer_frames_through_the_overlapped_gather_two_ranks_gloo(n_frames):
    """cfg3 as the product runs it, at toy size on CPU: 2 processes, each renders its round-robin share of the batch with the real
    Trainer and the round-r all_gather overlaps round r+1; every rank ends up with the whole batch, equal to a single-process
    render of the same frames (3 frames: the last round has an idle rank that contributes padding)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_trainer_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    render_one, S = _trainer_renderer()
    ref = torch.stack([render_one(k) for k in range(n_frames)], 0).numpy()
    assert np.abs(ref[0] - ref[1]).max() > 1e-4            # the frames of the batch really differ (head pose)
    assert np.array_equal(res[0][1], res[1][1])             # every rank holds the same batch, bit for bit
    for rank, out in res:                                   # (the single-process reference runs with another thread count: fp32 noise)
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 2e-5, rank


def test_shard_frames_partition():
    from havatar_amd.frames import shard_frames
    for n, w in ((64, 8), (7, 3), (1, 1), (3, 8)):
        parts = [shard_frames(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.parametrize("workload,ranks,frames", [("cfg3", 2, 3), ("cfg2", 8, 8), ("cfg3", 8, 11)])
def test_bench_multi_rank_branch_runs_under_gloo(workload, ranks, frames):
    """bench.py's N > 1 branch (process group, barrier-bracketed timing, MAX over ranks, rank 0 prints one JSON line; for cfg3 the
    round-by-round overlapped all_gather) executed for real under torch.distributed.run, CPU tensors, gloo: 2 ranks, and the 8 ranks of
    one MI355X node for both workloads (cfg3 with a ragged last round: 11 frames over 8 ranks).  `rccl_ranks_seen` is what the process
    group itself reports (one all-reduce over all ranks): the field the driver's SCALE record can check N against."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "0",
           "--device", "cpu", "--size", "8", "--workload", workload, "--frames", str(frames)]
    env = dict(os.environ, OMP_NUM_THREADS="1" if ranks > 2 else "2")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["ranks"] == ranks and d["steps"] == 1 and d["value"] > 0 and d["unit"] == "frames/s"
    seen = d["rccl_ranks_seen"]
    assert seen["ranks"] == ranks and seen["world_size"] == ranks and seen["backend"] == "gloo"
    assert seen["sum_of_local_device_indices"] == ranks * (ranks - 1) // 2          # every LOCAL_RANK 0..N-1 took part exactly once
    if workload == "cfg3":
        assert d["scaling"] == "strong" and d["config"]["frames_per_step"] == frames and d["config"]["gathered"] == [frames, 3, 8, 8]
    else:
        assert d["scaling"] == "weak" and d["config"]["frames_per_step"] == ranks
