"""Counterparts of the reference's entry scripts for the path: same CLI flags, checkpoint keys and
per-frame / per-step call sequence, running on the MI355X renderer."""
