import sys, os, glob
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from helpers import *
for f in sorted(glob.glob('tests/golden/render_*.npz')):
    name=os.path.basename(f)[:-4]
    g, sc, cfg, kw = load_render_fixture(name)
    o = hip_render(sc, **cfg, **kw)
    line=[]
    for k in OUT_KEYS:
        if 'ref_'+k in g.files:
            s=f"{k[:3]+k[-4:]}:{linf(o[k], g['ref_'+k]):.1e}"
            if 'ref64_'+k in g.files: s+=f"(v64 {linf(o[k], g['ref64_'+k]):.1e} nf {linf(g['ref_'+k], g['ref64_'+k]):.1e})"
            line.append(s)
    print(name[7:], ' '.join(line))
