#!/usr/bin/env python3
"""Kernel-only A/B of march-kernel builds on ONE box: python tools/ab_march.py libA.so libB.so ...  ('' = the in-tree library).
Each library is loaded in a fresh subprocess; prints median kernel ms over N launches for perturb on/off and a digest of the deterministic outputs."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from havatar_amd import synth
from havatar_amd.render import RayMarcher
dev = torch.device("cuda:0")
H = W = 512
sc = synth.scene(H, W, "primary")
rm = RayMarcher(sc["nerf_scale"], sc["nerf_trans"], sc["skin_scale"], sc["skin_trans"])
t = lambda a: torch.from_numpy(a).to(dev)
m = sc["mlp"]
rm.set_mlp(*[t(m[k]) for k in ("W1", "b1", "W2", "b2", "Wa", "ba", "Wf", "bf", "Wc", "bc")])
rm.set_triplane(t(sc["planes"]))
args = (t(sc["rays"]), t(sc["bg"]), t(sc["inv_T"]), t(sc["vol"]), 64, 16)
CO = os.environ.get('COARSE', '0') == '1'          # COARSE=1: all seven maps (the <., ., 0|1> kernels)
out = []
for perturb in (True, False):
    for _ in range(3): rm.render(*args, perturb=perturb, coarse_outputs=CO)
    ts = []
    for _ in range(int(os.environ.get('AB_N', '12'))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rm.render(*args, perturb=perturb, coarse_outputs=CO); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    out.append("%s %.3f ms (min %.3f)" % ("perturb" if perturb else "det    ", float(np.median(ts)), min(ts)))
import hashlib
o = rm.render(*args, perturb=False, coarse_outputs=CO)          # digest of the deterministic outputs: equal digests = bit-identical builds
o = o if isinstance(o, dict) else dict(enumerate(o))
hsh = hashlib.sha1()
for k in sorted(o, key=str):
    if torch.is_tensor(o[k]): hsh.update(o[k].detach().cpu().numpy().tobytes())
out.append("det sha1 " + hsh.hexdigest()[:12])
print(" | ".join(out))
'''
for rnd in range(int(os.environ.get('AB_ROUNDS', '2'))):          # AB_ROUNDS / AB_N: rounds over the libraries / timed launches per round
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib:
            env["HAVATAR_LIB"] = os.path.abspath(lib)
        else:
            env.pop("HAVATAR_LIB", None)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("%-50s %s" % (lib or "(in-tree)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
