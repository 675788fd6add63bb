"""hipGraph capture of a whole frame: `Trainer.forward` for fixed shapes becomes one graph launch.

A frame is ~250 kernel launches (two StyleGAN encoders through MIOpen + the HIP custom ops, the plane projection, the fused
ray march).  Eagerly, PyTorch's launch overhead is a third of the encoders' wall time; captured, the host cost is one
`hipGraphLaunch` + the input copies.  The captured ray march reads its RNG call counter from device memory and advances it
on the stream (HavRenderParams.rng_counter), so every replay draws fresh stratified jitter like the eager path.
"""
import torch


class GraphedForward:
    """Capture `module(**kwargs)` (tensors in `kwargs` are the per-frame inputs) and replay it with new inputs.

    Weights must not change between capture and replay (re-capture after a training step / load_state_dict)."""

    def __init__(self, module, example_kwargs, warmup=3):
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
