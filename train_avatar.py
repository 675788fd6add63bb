#!/usr/bin/env python3
"""Entry script with the reference's name and flags (train_avatar.py); see havatar_amd/harness/train.py."""
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")          # before the first HIP call (havatar_amd/__init__.py)

import numpy as np
import torch

from havatar_amd.harness.train import main

if __name__ == "__main__":
    np.random.seed(999)
    torch.random.manual_seed(999)
    main()
