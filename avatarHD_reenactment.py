#!/usr/bin/env python3
"""Entry script with the reference's name and flags (avatarHD_reenactment.py); see havatar_amd/harness/reenact.py."""
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")          # before the first HIP call (havatar_amd/__init__.py)

from havatar_amd.harness.reenact import main

if __name__ == "__main__":
    main()
