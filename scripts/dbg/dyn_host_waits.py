"""debug: where the REAL dynamic pipeline (DynamicPipeline.step, overlapped) waits: object-tracker rows, window solve, object solve (dv_timing_enable(ctx, -1))"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
N = 220
seq = DynamicSequence(1280, 720, sim.ZED, N + 1, rate=20.0, device="cuda:0", boxes=("escort", 4))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
for k in range(120): pipe.step()
gc.collect(); gc.freeze()
pipe.ctx.timing_enable(-1)
t0 = time.perf_counter()
for k in range(100): pipe.step()
dt = time.perf_counter() - t0
print("%.1f frames/s, %.1f us per frame" % (100 / dt, dt * 1e4))
for name in ("h_inst_wait", "h_solve_wait", "h_dyn_solve_wait", "h_process_begin", "h_dynamic", "h_process_end", "h_imu", "h_solve_begin"):
    ms, cnt = pipe.ctx.timing_get(name)
    if cnt: print("  %-18s %8.1f us  (n=%d)" % (name, 1e3 * ms / cnt, cnt))
pipe.ctx.close()
