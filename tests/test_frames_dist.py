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


def test_shard_frames_partition():
    from havatar_amd.frames import shard_frames
    for n, w in ((64, 8), (7, 3), (1, 1), (3, 8)):
        parts = [shard_frames(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
