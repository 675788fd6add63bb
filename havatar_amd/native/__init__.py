"""Native-surface modules: drop-ins for the reference's two pybind extensions, imported there BY BARE NAME
(`import fused`, `import upfirdn2d as upfirdn2d_op`; model/op/fused_act.py:20, model/op/upfirdn2d.py:19).

`havatar_amd.native.install()` registers them under those bare names so that the reference's own
model/op/*.py run unmodified on top of libhavatar_hip.so.
"""
import sys


def install():
    from . import fused, upfirdn2d
    sys.modules["fused"] = fused
    sys.modules["upfirdn2d"] = upfirdn2d
    return fused, upfirdn2d
