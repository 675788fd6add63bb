"""Counterparts of the reference's entry scripts for the path: same CLI flags, checkpoint keys and
per-frame / per-step call sequence, running on the MI355X renderer."""


def per_rank_miopen_cache():
    """One MIOpen user database / kernel cache per rank when several processes share a node: they all select solvers for the same
    convolution shapes at start-up and would otherwise queue on the locks of one sqlite file.  Call before the first convolution."""
    import os
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
        if var not in os.environ:
            d = os.path.join("/tmp", "havatar_miopen_rank%s" % os.environ.get("LOCAL_RANK", "0"), sub)
            os.makedirs(d, exist_ok=True)
            os.environ[var] = d
