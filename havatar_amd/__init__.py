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
# A caller that touches the device BEFORE importing this package (bench.py: set_device + init_process_group come first) must set it itself at the
# top of its script -- bench.py, train_avatar.py and avatarHD_reenactment.py do; hipgraph_state() tells which case a process is in.
HIPGRAPH_PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _hip_already_initialised():
    """Had this process made its first HIP call when the package was imported?  (Only torch can have done so on this path: if torch is not
    loaded yet, nothing has; `import torch` alone does not initialise the runtime, torch.cuda.set_device / a first tensor on the device does.)"""
    import sys as _sys
    _torch = _sys.modules.get("torch")
    try:
        return bool(_torch is not None and _torch.cuda.is_initialized())
    except Exception:
        return False


# state of the workaround, decided ONCE at import: the value the caller had chosen before (None = unset), whether HIP was already up
_ENV_BEFORE_IMPORT = _os.environ.get(HIPGRAPH_PACKET_CAPTURE_ENV)
_HIP_UP_AT_IMPORT = _hip_already_initialised()
_os.environ.setdefault(HIPGRAPH_PACKET_CAPTURE_ENV, "0")


def hipgraph_state():
    """What is known about the runtime's graph packet capture in this process:
    {"env": value now, "set_by": "caller" | "havatar_amd import", "hip_initialised_before_setting": bool, "in_force": bool}.
    `in_force` is True only if the variable was "0" BEFORE the runtime initialised: either the caller (bench.py, the harness entry
    scripts, the user's shell) had set it, or this package set it while HIP was still down."""
    env = _os.environ.get(HIPGRAPH_PACKET_CAPTURE_ENV)
    by_caller = _ENV_BEFORE_IMPORT is not None
    late = (not by_caller) and _HIP_UP_AT_IMPORT
    return {"env": env, "set_by": "caller" if by_caller else "havatar_amd import", "hip_initialised_before_setting": late,
            "in_force": env == "0" and not late}


def hipgraph_replays_safe():
    """True when this process runs the HIP runtime with the packet capture of graph launches off (see above): the variable is "0" AND it
    was set before the runtime's first HIP call.  A process that touched the device (torch.cuda.set_device, init_process_group, a first
    device tensor) before importing this package without setting the variable itself gets False -- and a warning from graph.py."""
    return hipgraph_state()["in_force"]
