"""debug: distribution of the per-frame period of the pipelined raw pipeline (no per-frame sync), in windows of 50 frames"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
n = 622
seq = SyntheticSequence(1280, 720, sim.ZED, n, rate=20.0, device="cuda:0")
pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10)
for _ in range(20):
    pipe.step()
torch.cuda.synchronize(); pipe.ctx.sync()
ts = [time.perf_counter()]
for k in range(n - 22):
    pipe.step(); ts.append(time.perf_counter())
d = np.diff(ts) * 1e3
for a in range(0, len(d) - 49, 50):
    w = d[a:a + 50]
    print(f"frames {a + 20:3d}-{a + 69:3d}: mean {w.mean():.3f} ms  p50 {np.median(w):.3f}  p95 {np.percentile(w, 95):.3f}  max {w.max():.3f}   -> {1e3 / w.mean():.0f} fps")
