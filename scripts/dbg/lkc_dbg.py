import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dynamic_vins_amd import synth
from dynamic_vins_amd.frontend import Context
from tests import oracle_py
o = oracle_py.load()
w, h, n = 192, 144, 60
ctx = Context(width=w, height=h, max_cnt=n, min_dist=8)
seq = synth.PlaneSequence(w, h, seed=11, disparity=2.0)
left, right = seq.frame(0)
pts = o.gftt(left, n, 0.01, 8, None)
rng = np.random.default_rng(1)
extra = np.array([[0.4, 0.4], [w - 1.2, h - 1.3], [w + 5.0, 3.0], [-2.0, -2.0], [w / 2 + 0.25, 0.6]], np.float32)
pts = np.concatenate([pts + rng.uniform(-0.45, 0.45, pts.shape).astype(np.float32), extra]).astype(np.float32)
for ml in (0, 1, 2, 3):
    po, so = o.lk_cuda(left, right, pts, ml, 30)
    pd, sd = ctx.lk_cuda(left, right, pts, ml, 30)
    bad = np.nonzero((po.view(np.uint32) != pd.view(np.uint32)).any(1) | (so != sd))[0]
    print("ml", ml, "mismatch", len(bad), bad[:10])
    for i in bad[:4]:
        print("   ", i, pts[i], po[i], pd[i], so[i], sd[i])
for it in (0, 1, 2, 3):
    po, so = o.lk_cuda(left, right, pts, 0, it)
    pd, sd = ctx.lk_cuda(left, right, pts, 0, it)
    bad = np.nonzero((po.view(np.uint32) != pd.view(np.uint32)).any(1) | (so != sd))[0]
    print("iters", it, "mismatch", len(bad), bad[:10])
    for i in bad[:3]:
        print("   ", i, pts[i], po[i], pd[i], so[i], sd[i])
