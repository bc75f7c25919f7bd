"""debug: per-frame wall time, iterations and slots of the raw pipeline (20-frame averages)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
n = 262
seq = SyntheticSequence(1280, 720, sim.ZED, n, rate=20.0, device="cuda:0")
pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10)
if len(sys.argv) > 1:
    assert pipe.ctx.lib.dv_debug_set(pipe.ctx.h, sys.argv[1].encode(), 1) == 0
ts, its, sl, nl = [], [], [], []
for k in range(n - 2):
    pipe.ctx.sync(); t0 = time.perf_counter()
    st = pipe.step()
    pipe.ctx.sync(); ts.append(time.perf_counter() - t0); its.append(st.iterations); nl.append(st.n_long)
    sl.append(pipe.est.last_summary_slots() if hasattr(pipe.est, "last_summary_slots") else 0)
ts = np.array(ts) * 1e3
for a in range(20, n - 2, 20):
    print(f"frames {a:3d}-{a + 19:3d}: {ts[a:a + 20].mean():.3f} ms  max {ts[a:a + 20].max():.3f}  iterations {np.mean(its[a:a + 20]):.2f}  landmarks {np.mean(nl[a:a + 20]):.0f}")
