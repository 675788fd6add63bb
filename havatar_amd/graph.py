"""hipGraph capture of a whole frame: `Trainer.forward` for fixed shapes becomes one graph launch.

A frame is ~250 kernel launches (two StyleGAN encoders through MIOpen + the HIP custom ops, the plane projection, the fused
ray march).  Eagerly, PyTorch's launch overhead is a third of the encoders' wall time; captured, the host cost is one
`hipGraphLaunch` + the input copies.  The captured ray march reads its RNG call counter from device memory and advances it
on the stream (HavRenderParams.rng_counter), so every replay draws fresh stratified jitter like the eager path.
"""
import os

import torch

_WEIGHTS_EPOCH = [0]
_WARNED = [False]


def _check_runtime_workaround():
    """Graph replays interleaved with eager launches need DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 on this ROCm stack (havatar_amd/__init__.py sets
    it at import unless the caller chose otherwise).  A process that initialised HIP before importing this package, or that switched the
    packet capture back on, is told once."""
    import warnings
    from . import HIPGRAPH_PACKET_CAPTURE_ENV, hipgraph_replays_safe, hipgraph_state
    if not hipgraph_replays_safe() and not _WARNED[0]:
        _WARNED[0] = True
        st = hipgraph_state()
        why = ("the HIP runtime was initialised before havatar_amd was imported and %s was not set: set it to 0 at the top of the script, before "
               "the first torch.cuda call" % HIPGRAPH_PACKET_CAPTURE_ENV) if st["hip_initialised_before_setting"] else \
              ("%s = %r, not 0" % (HIPGRAPH_PACKET_CAPTURE_ENV, st["env"]))
        warnings.warn("hipGraph packet capture may be on (%s): on ROCm 7.0.x replays of a captured hipGraph can return stale results once a "
                      "reduction kernel has been launched eagerly between replays (tools/repro_graph_reduce.py)" % why, RuntimeWarning)


def weights_epoch():
    """Counter folded into every weight-derived cache key (packed MLP blob, weight-only encoder caches): parameters updated by
    a replayed graph do not bump `Tensor._version`, so GraphedTrainStep bumps this instead."""
    return _WEIGHTS_EPOCH[0]


def bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


class GraphedForward:
    """Capture `module(**kwargs)` (tensors in `kwargs` are the per-frame inputs) and replay it with new inputs.

    Weights must not change between capture and replay (re-capture after a training step / load_state_dict)."""

    def __init__(self, module, example_kwargs, warmup=3):
        _check_runtime_workaround()
        self.module = module
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_kwargs.items()}
        dev = next(v.device for v in self.static.values() if torch.is_tensor(v))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # MIOpen solver selection, lazy buffers, weight packing
                module(**self.static)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = module(**self.static)

    def __call__(self, **kwargs):
        for k, v in kwargs.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
            elif self.static.get(k) != v:
                raise RuntimeError(f"GraphedForward: non-tensor argument {k!r} changed since capture")
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """One optimisation step -- forward, backward, optimiser update -- as ONE hipGraph launch.

    cfg5's step is ~3 800 kernel launches for ~55 ms of kernel time: eagerly the host (Python autograd + launch overhead) is the
    bottleneck at ~80 ms per step.  `step_fn(**tensors) -> (loss, aux)` must be capture-safe (no .item(), no host tensors: the
    Trainer draws sample_pdf's stratified u on the device while capturing) and the optimiser must be `capturable`.  Run at least
    one eager step first (solver selection, lazy state); capturing itself executes nothing, so no step is spent on it."""

    def __init__(self, step_fn, optimizer, example, stream=None):
        # stream: capture on the stream the eager warm-up steps ran on.  The parameters' AccumulateGrad nodes are bound to the stream
        # they were created on and can outlive an iteration; captured on another stream, the backward pass becomes a graph with side
        # branches (PyTorch warns: "AccumulateGrad node's stream does not match ... may break CUDA graph capture"), and on this stack
        # replays of such a graph were seen to run consecutive kernels of the main chain out of order (DESIGN.md 7,
        # profiles/r04_flake_bisect.txt): intermittent non-finite activations, gone with AMD_SERIALIZE_KERNEL=3 and with this.
        _check_runtime_workaround()
        self.static = {k: v.clone() for k, v in example.items()}
        self.optimizer = optimizer
        optimizer.zero_grad(set_to_none=True)
        shape = os.environ.get("HAVATAR_GRAPH_SHAPE") == "1"          # development aid: print the captured graph's topology (tools/graph_shape.py)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if shape else torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=stream, capture_error_mode="thread_local"):
            self.loss, self.aux = step_fn(**self.static)
            self.loss.backward()
            optimizer.step()
        if shape:
            import importlib.util
            spec = importlib.util.spec_from_file_location("graph_shape", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "graph_shape.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            print("GraphedTrainStep: captured graph", mod.describe(self.graph.raw_cuda_graph()), flush=True)
            self.graph.instantiate()

    def matches(self, tensors):
        return tensors.keys() == self.static.keys() and all(tensors[k].shape == v.shape and tensors[k].dtype == v.dtype
                                                            for k, v in self.static.items())

    def __call__(self, **tensors):
        for k, v in tensors.items():
            self.static[k].copy_(v, non_blocking=True)
        from .native import conv as _nconv
        if _nconv._NAN_TRACE is not None:          # development aid: the replay's is-finite flags start from zero
            _nconv.nan_trace_reset(next(iter(self.static.values())).device)
        self.graph.replay()
        bump_weights_epoch()
        return self.loss, self.aux
