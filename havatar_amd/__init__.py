"""havatar_amd -- MI355X (gfx950) implementation of HAvatar's volumetric-rendering hot path.

Only what that path needs lives here: `csrc/` (HIP kernels + the C ABI of include/havatar.h), the ctypes
binding (`_lib`), the torch-level host wrappers (`render`, `native.fused`, `native.upfirdn2d`) and the mirror of the
reference's Python surface (`model/`, `utils/`).  There is no CPU fallback for CUDA/HIP tensors: if
libhavatar_hip.so is missing the product path raises.
"""
__version__ = "0.1.0"

import os as _os

# ROCm runtime fault (HIP 7.0.x, clr "graph packet capture"), root-caused in round 5 (DESIGN.md "graphed training step"; stand-alone reproducer
# without this package: tools/repro_graph_reduce.py): once ANY reduction kernel has been launched eagerly between two replays of a captured
# hipGraph, some kernel nodes of later replays stop producing their results (a captured torch.sum of ones returns the stale bytes of its output
# block, is-finite reductions read false on finite tensors) -- every replay from then on.  With the packets built at launch time instead
# (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0) the fault is gone in the reproducer (0 of 180 replays vs 56 of 60) and in the graphed training step (0
# flagged replays of 1 460 under the tracers that showed 31 of 83), at no measurable cost (frame 9.04 vs 9.06 ms, training step 24.3 vs 24.1 ms).
# The runtime reads the variable when it initialises (the first HIP call of the process), so it is set here, at import; an explicit setting wins.
HIPGRAPH_PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_os.environ.setdefault(HIPGRAPH_PACKET_CAPTURE_ENV, "0")


def hipgraph_replays_safe():
    """True when this process runs the HIP runtime with the packet capture of graph launches off (see above)."""
    return _os.environ.get(HIPGRAPH_PACKET_CAPTURE_ENV) == "0"
