"""Build libhavatar_hip.so (gfx950) in-tree: `python -m havatar_amd.build`.

One `hipcc --offload-arch=gfx950 -shared` per source set; hipcc cross-compiles without a GPU.  The .so is
git-ignored but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhavatar_hip.so")
SOURCES = ["hav_ops.hip", "hav_render.hip", "hav_train.hip", "hav_mlp_train.hip", "hav_conv.hip"]
HEADERS = ["hav_common.h", os.path.join("..", "..", "include", "havatar.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
BUILD_INFO = os.path.join(LIBDIR, "BUILD_INFO.json")
# The kernels were validated -- parity suite, full-occupancy determinism stress (tools/stress_production.py, tools/stress_diag.py: a
# rare run-to-run difference turned out to depend on the compiler's instruction selection, docs/history/DESIGN_r1-r4.md 3.12) -- with exactly this
# compiler.  A build with another one is recorded as untested in lib/BUILD_INFO.json and warned about; it is refused only when
# HAVATAR_REQUIRE_TESTED_HIPCC=1.  (The .so built here with hipcc 7.2 runs on the GPU boxes' ROCm 7.0.2 runtime / HIP 7.0.51831 --
# code objects are forward-compatible there; GPUTEST_r0*.json record that pairing.)
TESTED_HIPCC = ("HIP version: 7.2.26015-fc0010cf6a",
                "AMD clang version 22.0.0git (https://github.com/RadeonOpenCompute/llvm-project roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5)")


def hipcc_version(hipcc):
    out = subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=False).stdout.decode(errors="replace")
    return tuple(ln.strip() for ln in out.splitlines()[:2])


def check_compiler(hipcc):
    """Identity of the compiler that is about to build the library: (version lines, tested?).  An unvalidated compiler is a warning
    (and `tested: false` in BUILD_INFO.json); with HAVATAR_REQUIRE_TESTED_HIPCC=1 it is an error."""
    ver = hipcc_version(hipcc)
    tested = ver == TESTED_HIPCC
    if not tested:
        msg = ("hipcc differs from the compiler this library's kernels were validated with:\n  found  %s\n  tested %s\n"
               "Re-run tools/stress_production.py and the -m gpu determinism tests on a GPU with the new build." % (ver, TESTED_HIPCC))
        if os.environ.get("HAVATAR_REQUIRE_TESTED_HIPCC", "0") == "1":
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return ver, tested


def _stale():
    """Rebuild only when the library is missing or older than a source: an existing, working .so is never rebuilt (or refused) because of
    the compiler that happens to be installed or because BUILD_INFO.json is absent."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_info():
    """What lib/BUILD_INFO.json says about the library on disk (None if it was not written by build())."""
    import json
    try:
        return json.load(open(BUILD_INFO))
    except (OSError, ValueError):
        return None


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libhavatar_hip.so. Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    ver, tested = check_compiler(hipcc)          # only when a build is actually needed
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    import json
    with open(BUILD_INFO, "w") as f:          # travels with the .so: build_info() reads it, tests/test_abi.py checks it
        json.dump({"hipcc": list(ver), "tested": tested, "flags": FLAGS, "sources": SOURCES}, f, indent=1)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
