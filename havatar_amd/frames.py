"""Frame sharding for batch-split inference (BASELINE config 3).

Frames are independent units of the hot path: rank r of N renders frames r, r+N, ... with a full replica of the weights;
there is NO collective inside the path.  The only exchange is the optional gather of finished frames (one all_gather per
batch, RCCL over xGMI on GPUs -- `backend="nccl"` IS RCCL on ROCm -- gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world_size):
    """Indices of the frames this rank renders (round-robin, so a 64-frame batch gives 8 per GPU on 8 GPUs)."""
    return list(range(rank, n_frames, world_size))


def gather_frames(local, n_frames, group=None):
    """local: [n_local, ...] frames rendered by this rank (shard_frames order) -> [n_frames, ...] on every rank.

    One all_gather of equal-sized (zero-padded) shards; on a full-mesh xGMI node each GPU pushes its shard down all links
    concurrently (direct algorithm), 25 MB per rank for 8 fp32 512x512x3 frames."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_frames + world - 1) // world
    pad = local.new_zeros((per,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = local.new_empty((n_frames,) + tuple(local.shape[1:]))
    for r in range(world):
        idx = shard_frames(n_frames, r, world)
        out[idx] = bufs[r][: len(idx)]
    return out


def render_frame_batch(render_one, n_frames, group=None, gather=True):
    """Render frames [0, n_frames) sharded over the process group. `render_one(k)` -> tensor for frame k (same shape for all k).
    Returns [n_frames, ...] on every rank if gather else this rank's [n_local, ...]."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = [render_one(k) for k in shard_frames(n_frames, rank, world)]
    local = torch.stack(mine, 0) if mine else None
    if local is None:
        raise RuntimeError("more ranks than frames")
    return gather_frames(local, n_frames, group) if gather else local


class OverlappedFrameGather:
    """The batch-split inference path's ONE exchange step: finished frames are all-gathered round by round --
    round r = frame r * world + rank on every rank -- and the collective of round r runs WHILE round r + 1 is being rendered.

    On HIP the payload is snapshotted on the render stream (the renderer's output buffer is static under a hipGraph and the next
    replay overwrites it), the collective is issued from a side stream (RCCL then runs it on its own stream, ordered after the
    snapshot), and nothing on the host waits until `finalize()`.  3.1 MB per rank and round for a 512x512 RGB fp32 frame: on the xGMI
    full mesh every GPU pushes its frame down all 7 links at once (direct all-gather), ~20 us against >= 9 ms of rendering.
    On CPU tensors (gloo, the tests) the same calls run with `async_op=True` in gloo's worker thread.
    With world_size 1 (or no process group) it only collects the frames -- unless `force_collective` is set and a process group
    exists: then a 1-rank group goes through the very same snapshot / side-stream / all_gather_into_tensor / stream-wait sequence
    (how the HIP + RCCL branch is exercised on a single GPU: tests/test_frames_gpu.py, bench.py --workload cfg3 --force-collective 1)."""

    def __init__(self, n_frames, frame_shape, dtype=torch.float32, device="cpu", group=None, force_collective=False):
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.collective = on and (force_collective or dist.get_world_size(group) > 1)
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.n_frames = n_frames
        self.rounds = (n_frames + self.world - 1) // self.world
        self.device = torch.device(device)
        self.out = torch.zeros((self.rounds, self.world) + tuple(frame_shape), dtype=dtype, device=self.device)
        # [1, ...]: all_gather_into_tensor concatenates along dim 0 (gloo insists on it; RCCL accepts it)
        self.stage = [torch.zeros((1,) + tuple(frame_shape), dtype=dtype, device=self.device) for _ in range(2)]
        self.works = [None] * self.rounds
        self.side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def my_frame(self, r):
        """Frame index this rank renders in round r (None: the batch has run out, the rank contributes a zero frame)."""
        k = r * self.world + self.rank
        return k if k < self.n_frames else None

    def submit(self, r, frame):
        """frame: this rank's finished frame of round r (or None), produced on the current stream."""
        if r >= 2 and self.works[r - 2] is not None:
            self.works[r - 2].wait()                 # the staging slot is free again (a stream-side wait on HIP, not a host wait)
        st = self.stage[r & 1]
        if frame is None:
            st.zero_()
        else:
            st.copy_(frame.reshape(st.shape), non_blocking=True)
        if not self.collective:
            self.out[r].copy_(st, non_blocking=True)
            return
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.side):
                self.works[r] = dist.all_gather_into_tensor(self.out[r], st, group=self.group, async_op=True)
        else:
            self.works[r] = dist.all_gather_into_tensor(self.out[r], st, group=self.group, async_op=True)

    def finalize(self):
        """-> [n_frames, ...] in frame order on every rank (round-robin sharding makes [round][rank] the frame order)."""
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = [None] * self.rounds
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return self.out.reshape((self.rounds * self.world,) + tuple(self.out.shape[2:]))[: self.n_frames]
