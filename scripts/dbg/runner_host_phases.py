"""debug: host phase times (h_*) of the raw workload driven by the C++ runner (dv_runner), dv_timing_enable(ctx, -1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
N = 150
seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
pipe = Pipeline(seq)
r = Runner([pipe], group_size=0, threads=1)
r.run(40)
pipe.ctx.timing_enable(-1)
torch.cuda.synchronize(); t0 = time.perf_counter(); r.run(100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%.1f frames/s, %.1f us per frame" % (100 / dt, dt * 1e4))
tot = 0.0
for name in ("h_imu", "h_add_features", "h_triangulate", "h_build", "h_solve_begin", "h_solve_upload", "h_solve_enqueue", "h_solve_wait", "h_post", "h_reject", "h_slide", "h_process_begin", "h_process_end"):
    ms, cnt = pipe.ctx.timing_get(name)
    if cnt: print("  %-18s %8.1f us  (n=%d)" % (name, 1e3 * ms / cnt, cnt))
r.close(); pipe.ctx.close()
