"""debug: per-frame wall time of the dynamic pipeline (bench workload); prints the slow frames"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
N = 232
seq = DynamicSequence(1280, 720, sim.ZED, N + 1, rate=20.0, device="cuda:0", boxes=("escort", 4))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
ts = []
for k in range(N):
    if k == 20: gc.collect(); gc.freeze()
    t0 = time.perf_counter(); pipe.step(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("p50 %.2f p95 %.2f max %.2f ms" % (np.percentile(ts[20:], 50), np.percentile(ts[20:], 95), ts[20:].max()))
print("frames over 2.5 ms:", [(i, round(float(t), 2)) for i, t in enumerate(ts) if t > 2.5 and i >= 12])
for a in range(20, N, 20): print(a, "%.2f" % ts[a:a + 20].mean(), end="  ")
print()
pipe.ctx.close()
