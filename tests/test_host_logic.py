"""Host-side / formula-level checks that need neither the GPU nor the oracle."""
import numpy as np


def _fmix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x85EBCA6B); x ^= x >> np.uint32(13); x *= np.uint32(0xC2B2AE35); x ^= x >> np.uint32(16)
    return x


def jitter_uniform(gr, s, stream, seed=0x1234ABCD5678EF01, call=0):
    """NumPy restatement of rng_ray_key / rng_uniform in havatar_amd/csrc/hav_render.hip (the stratified-jitter stream)."""
    gr = np.asarray(gr, np.uint64)
    with np.errstate(over="ignore"):
        key = _fmix32((gr & np.uint64(0xFFFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779B1) + np.uint32(seed & 0xFFFFFFFF)) \
            ^ ((gr >> np.uint64(32)).astype(np.uint32) * np.uint32(0x7FEB352D))
        call_key = np.uint32(((seed >> 32) + call * 0x68E31DA4) & 0xFFFFFFFF)
        word = np.asarray(s, np.uint32) * np.uint32(0x846CA68B) + np.uint32((stream * 0x632BE5AB) & 0xFFFFFFFF) + call_key
        x = _fmix32(key ^ word)
    return (x >> np.uint32(8)).astype(np.float64) / 16777216.0


def test_jitter_stream_is_equidistributed_and_uncorrelated():
    """The block kernel draws one uniform per (ray, sample) from this counter hash: check what stratified sampling needs --
    flat histogram, right moments, no correlation along samples, along rays, between streams or between calls."""
    rays = np.arange(4096, dtype=np.uint64)[:, None] + np.uint64(3 * 262144)
    samp = np.arange(64, dtype=np.uint32)[None, :]
    u = jitter_uniform(rays, samp, 1)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1.0 / 12.0) < 1e-3
    hist = np.histogram(u, bins=64, range=(0, 1))[0]
    chi2 = ((hist - u.size / 64) ** 2 / (u.size / 64)).sum()
    assert chi2 < 120.0                                   # 63 dof: mean 63, p(chi2 > 120) ~ 1e-5
    c = lambda a, b: abs(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    assert c(u[:, :-1], u[:, 1:]) < 0.01                  # neighbouring samples of a ray
    assert c(u[:-1], u[1:]) < 0.01                        # neighbouring rays (pixels) at the same sample
    assert c(u, jitter_uniform(rays, samp, 2)) < 0.01     # xi vs zeta streams
    assert c(u, jitter_uniform(rays, samp, 1, call=1)) < 0.01      # consecutive frames (device call counter)
    # per-sample means over rays (what a pixel neighbourhood sees) and per-ray means over samples
    assert np.abs(u.mean(0) - 0.5).max() < 0.03 and np.abs(u.mean(1) - 0.5).max() < 0.2
    assert len(np.unique(u)) > 0.98 * u.size


def test_frame_sharding_is_a_partition():
    from havatar_amd.frames import shard_frames
    for n in (0, 1, 7, 64):
        for world in (1, 2, 8):
            parts = [list(shard_frames(n, r, world)) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_hipgraph_workaround_state_tells_a_late_import_from_an_early_one():
    """DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 only helps if it is set before the process's first HIP call.  havatar_amd.hipgraph_state()
    must say `in_force: False` when the package is imported after the device was touched and nobody had set the variable (fresh interpreters;
    the "device was touched" case is simulated with a stand-in `torch` module whose cuda.is_initialized() returns True)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, types, json, os\n"
            "sys.path.insert(0, %r)\n"
            "if os.environ.get('FAKE_HIP_UP') == '1':\n"
            "    t = types.ModuleType('torch'); t.cuda = types.SimpleNamespace(is_initialized=lambda: True); sys.modules['torch'] = t\n"
            "import havatar_amd\n"
            "print(json.dumps([havatar_amd.hipgraph_state(), havatar_amd.hipgraph_replays_safe()]))\n") % root

    def run(env_value, hip_up):
        env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
        if env_value is not None:
            env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = env_value
        env["FAKE_HIP_UP"] = "1" if hip_up else "0"
        out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, check=True).stdout
        return json.loads(out.strip().splitlines()[-1])

    st, safe = run(None, False)          # the normal import: nothing set, HIP still down -> the package's own setting is in force
    assert st == {"env": "0", "set_by": "havatar_amd import", "hip_initialised_before_setting": False, "in_force": True} and safe
    st, safe = run(None, True)           # device touched first, variable unset: the setdefault came too late
    assert st["env"] == "0" and st["hip_initialised_before_setting"] and not st["in_force"] and not safe
    st, safe = run("0", True)            # the caller set it at the top of its script (bench.py, the entry scripts): fine whatever came next
    assert st["set_by"] == "caller" and st["in_force"] and safe
    st, safe = run("1", False)           # an explicit setting wins and is reported as what it is
    assert st["env"] == "1" and not st["in_force"] and not safe


def test_entry_scripts_set_the_workaround_before_importing_torch():
    """bench.py / train_avatar.py / avatarHD_reenactment.py touch the device before (or while) importing the package: the variable must be set
    in their first statements, ahead of any `import torch`."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("bench.py", "train_avatar.py", "avatarHD_reenactment.py"):
        src = open(os.path.join(root, name)).read()
        setpos = src.index('os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")')
        first_torch = re.search(r"^\s*(import torch|from torch|from havatar_amd|import havatar_amd)", src, re.M)
        assert first_torch is not None and setpos < first_torch.start(), name
