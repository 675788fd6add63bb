#!/usr/bin/env python3
"""The radiance MLP's training kernels (hav_mlp_train_*) alone at config 5's size (0.92 M queries per step as two passes): forward and
backward time, rate against the bf16 matrix peak.  HAVATAR_LIB=<alternative .so> for A/B builds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.native import mlp_train
dev = torch.device("cuda:0")
NAMES = ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")
m = synth.scene(4, 4, "primary")["mlp"]
ws = [torch.from_numpy(np.ascontiguousarray(m[k])).to(dev) for k in NAMES]


class _M:
    def mlp_tensors(self):
        return ws


q = 917504
r = mlp_train.bench_kernels(_M(), q, dev, reps=9)
fl = 94848.0 * q
r.update({"queries": q, "fwd_TFLOPs": round(fl / r["fwd_ms"] / 1e9, 1), "bwd_TFLOPs": round(2 * fl / r["bwd_ms"] / 1e9, 1),
          "fwd_plus_bwd_frac_of_bf16_peak": round(3 * fl / (r["fwd_ms"] + r["bwd_ms"]) / 1e9 / 2500, 4)})
print(json.dumps(r))
