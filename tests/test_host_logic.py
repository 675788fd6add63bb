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
