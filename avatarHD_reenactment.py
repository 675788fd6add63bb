#!/usr/bin/env python3
"""Entry script with the reference's name and flags (avatarHD_reenactment.py); see havatar_amd/harness/reenact.py."""
from havatar_amd.harness.reenact import main

if __name__ == "__main__":
    main()
