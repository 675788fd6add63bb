"""GPU: the exchange step of the batch-split inference path (BASELINE config 3) on HIP tensors.

A 1-rank RCCL group is enough to run `OverlappedFrameGather`'s HIP branch for real -- snapshot of the hipGraph's static output on the
render stream, all_gather_into_tensor issued from a side stream, double-buffered staging with stream-side waits, the host only
waiting in finalize() -- and what it must deliver is checkable: the gathered batch is the sequentially rendered batch.  A late snapshot would return the
NEXT frame, a missing stream wait a half-written one; gloo on CPU cannot see either.  Two renderers: a captured toy graph with a
static output buffer (bit-exact check) and the real Trainer under GraphedForward (its encoders are not bit-reproducible from replay
to replay -- 3e-5 -- so that one is held to 2e-4 against frames that differ by > 1e-2 from their neighbours)."""
import os
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def nccl_one_rank():
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))       # "nccl" IS RCCL on ROCm
    yield dist
    dist.destroy_process_group()


def _trainer(S, dev):
    from havatar_amd import synth
    from havatar_amd.model.nerf_trainer import Trainer
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = CfgNode(synth.harness_config(render_size=S, gen_size=4 * S, img_res=S))
    v = cfg.nerf.validation
    v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = 64, 16, False, 0.0      # deterministic depths: frames are reproducible
    torch.manual_seed(0)
    tr = synth.fill_state_dict(Trainer(cfg, 1).requires_grad_(False)).eval().to(dev)
    tr.headpose_skin_net.fix_canonical_W()
    t = lambda a: torch.from_numpy(a).to(dev)
    front, left, right = [t(a) for a in synth.cond_images()]
    data = dict(ray_batch=t(synth.camera_rays(S, S))[None], background_prior=torch.ones(1, S * S, 3, device=dev),
                inv_head_T=t(synth.frame_pose(0))[None], front_render_cond=front, left_render_cond=left, right_render_cond=right,
                mode="validation", fidx=0, render_full_img=True)
    return tr, data, (lambda k: t(synth.frame_pose(k))[None])


@pytest.mark.parametrize("n_frames", [5])
def test_overlapped_gather_hip_branch_with_the_real_trainer_under_a_hipgraph(nccl_one_rank, n_frames):
    from havatar_amd.frames import OverlappedFrameGather
    from havatar_amd.graph import GraphedForward
    dev = torch.device("cuda", 0)
    S = 64
    tr, data, pose = _trainer(S, dev)
    frame = GraphedForward(tr, data)
    # reference: one frame at a time, cloned before the next replay overwrites the graph's static output
    ref = []
    for k in range(n_frames):
        out = frame(inv_head_T=pose(k))[0][0, :3]
        torch.cuda.synchronize()
        ref.append(out.clone())
    ref = torch.stack(ref, 0)
    for k in range(n_frames - 1):
        assert (ref[k] - ref[k + 1]).abs().max() > 1e-2                # neighbouring frames really differ (head pose)
    g = OverlappedFrameGather(n_frames, (3, S, S), device=dev, force_collective=True)
    assert g.collective and g.side is not None and g.world == 1
    for rep in range(2):                                               # twice: the second batch re-uses staging slots and work handles
        for r in range(g.rounds):                                      # no host synchronisation anywhere in the loop
            g.submit(r, frame(inv_head_T=pose(g.my_frame(r)))[0][0, :3])
        got = g.finalize()
        torch.cuda.synchronize()
        assert got.shape == ref.shape
        err = (got - ref).abs().amax(dim=(1, 2, 3))
        assert float(err.max()) <= 2e-4, "batch %d: gathered frames differ from the sequential renders: %s" % (rep, err.tolist())


def test_overlapped_gather_hip_branch_is_bit_exact_on_a_captured_static_output(nccl_one_rank):
    """The ordering hazards in isolation, bit for bit: a captured graph writes frame k (a long dependent chain, so that it is still
    running when the host returns) into ONE static buffer; 24 rounds through the side-stream all_gather without any host wait."""
    from havatar_amd.frames import OverlappedFrameGather
    dev = torch.device("cuda", 0)
    n, shape = 24, (3, 256, 256)
    seed = torch.zeros(1, device=dev)
    static_out = torch.empty(shape, device=dev)
    base = torch.randn(shape, device=dev)

    def body():
        x = base + seed
        for _ in range(40):                                            # ~40 dependent elementwise kernels
            x = torch.sin(x) + 0.5 * x
        static_out.copy_(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()

    def render(k):
        seed.fill_(float(k))
        graph.replay()
        return static_out
    ref = []
    for k in range(n):
        render(k); torch.cuda.synchronize(); ref.append(static_out.clone())
    ref = torch.stack(ref, 0)
    g = OverlappedFrameGather(n, shape, device=dev, force_collective=True)
    for rep in range(3):
        for r in range(g.rounds):
            g.submit(r, render(g.my_frame(r)))
        got = g.finalize()
        torch.cuda.synchronize()
        assert torch.equal(got, ref), "batch %d: %s" % (rep, [(k, int(torch.equal(got[k], ref[k]))) for k in range(n)])


def test_gather_frames_one_rank_rccl(nccl_one_rank):
    """frames.gather_frames / all_gather_into_tensor on device tensors through the 1-rank RCCL group (a forced collective)."""
    from havatar_amd.frames import OverlappedFrameGather
    dev = torch.device("cuda", 0)
    g = OverlappedFrameGather(3, (4, 8), device=dev, force_collective=True)
    xs = [torch.randn(4, 8, device=dev) for _ in range(3)]
    for r, x in enumerate(xs):
        g.submit(r, x)
    assert torch.equal(g.finalize(), torch.stack(xs, 0))


@pytest.mark.parametrize("B", [2, 4])
def test_batched_frames_equal_the_frames_rendered_one_at_a_time(B):
    """Throughput mode of BASELINE config 3 (bench.py --workload cfg3 --frame-batch B, HAVATAR_FRAME_BATCH of the reenactment CLI): B frames
    per call -- the encoders (model/nerf_model.py:58-86 take a batch) see B condition sets, the march B x R rays in one launch.  Every frame
    has its own pose AND its own condition images.  (1) The march itself is bit-exact: given the planes of the one-frame calls, frame b of
    the batched launch equals the one-frame launch.  (2) The whole frame: the encoders' kernels pick other tilings at B > 1 (their
    results move by rounding, <= 3e-5 on the planes), the rendered frame stays within 1e-3 of the one rendered alone (measured 1e-4)."""
    dev = torch.device("cuda", 0)
    S = 64
    tr, data, pose = _trainer(S, dev)
    scale = lambda k: 1.0 - 0.07 * k
    one = lambda k: {**data, "inv_head_T": pose(k), **{c: data[c] * scale(k) for c in ("front_render_cond", "left_render_cond", "right_render_cond")}}
    singles, planes = [], []
    with torch.no_grad():
        for k in range(B):
            render, mask, _ = tr(**one(k))
            singles.append((render.clone(), mask.clone()))
            planes.append(tr.model_coarse.triPlane_embeddings.clone())
        rep = lambda x: x.expand(B, *x.shape[1:]).contiguous()
        batch = {**data, "ray_batch": rep(data["ray_batch"]), "background_prior": rep(data["background_prior"]),
                 "inv_head_T": torch.cat([pose(k) for k in range(B)]),
                 **{c: torch.cat([data[c] * scale(k) for k in range(B)]) for c in ("front_render_cond", "left_render_cond", "right_render_cond")}}
        render_b, mask_b, _ = tr(**batch)
        planes_b = tr.model_coarse.triPlane_embeddings.clone()
    assert render_b.shape == (B,) + tuple(singles[0][0].shape[1:])
    for k in range(B):
        assert float((planes_b[:, k] - planes[k][:, 0]).abs().max()) <= 3e-5 * float(planes[k].abs().max()) + 1e-6
        assert float((render_b[k:k + 1] - singles[k][0]).abs().max()) <= 1e-3
        assert float((mask_b[k:k + 1] - singles[k][1]).abs().max()) <= 1e-3
        assert float((singles[k][0] - singles[(k + 1) % B][0]).abs().max()) > 1e-2          # the frames really differ
    # (1) the march alone, on the one-frame planes stacked into a batch
    m = tr._hip_marcher()
    rd = batch["ray_batch"][..., 3:6]
    rays = torch.cat((batch["ray_batch"], rd / rd.norm(p=2, dim=-1).unsqueeze(-1)), dim=-1)
    vol = tr.headpose_skin_net.current_volume().detach()
    with torch.no_grad():
        m.set_mlp(*[x.detach() for x in tr.model_coarse.mlp_tensors()])
        m.set_triplane(torch.cat(planes, dim=1))
        out_b = [o.clone() if o is not None else None for o in m.render(rays, batch["background_prior"], batch["inv_head_T"], vol, 64, 16, perturb=False)]
        for k in range(B):
            m.set_triplane(planes[k])
            out_1 = m.render(rays[k:k + 1], batch["background_prior"][k:k + 1], batch["inv_head_T"][k:k + 1], vol, 64, 16, perturb=False)
            for a, b in zip(out_b, out_1):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a[k:k + 1], b)
