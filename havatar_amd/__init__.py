"""havatar_amd -- MI355X (gfx950) implementation of HAvatar's volumetric-rendering hot path.

Only what that path needs lives here: `csrc/` (HIP kernels + the C ABI of include/havatar.h), the ctypes
binding (`_lib`), the torch-level host wrappers (`render`, `native.fused`, `native.upfirdn2d`) and the mirror of the
reference's Python surface (`model/`, `utils/`).  There is no CPU fallback for CUDA/HIP tensors: if
libhavatar_hip.so is missing the product path raises.
"""
__version__ = "0.1.0"
