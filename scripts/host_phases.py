"""Host wall-clock per phase of the estimator (the "h_*" scopes of est_host.hip / be_api.hip) with the pipeline running overlapped, on the
benchmark's workload (dv_timing_enable(ctx, -1): no events, no extra synchronisation).  Shows what stands between the state download of
frame k and the upload of frame k+1 — the part of the frame the BA stream spends waiting for the host."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamic_vins_amd import sim                                     # noqa: E402
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence    # noqa: E402

W, H, STEPS, WARM = 1280, 720, 100, 20
seq = SyntheticSequence(W, H, sim.ZED, WARM + STEPS + 1, rate=20.0, phase=0.0, device="cuda:0")
pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0)
for _ in range(WARM):
    pipe.step()
pipe.ctx.timing_enable(-1)
pipe.ctx.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    pipe.step()
pipe.ctx.sync()
dt = time.perf_counter() - t0
print("%.1f frames/s, %.1f us per frame" % (STEPS / dt, dt / STEPS * 1e6))
for name in ("h_imu", "h_add_features", "h_triangulate", "h_build", "h_solve_begin", "h_solve_upload", "h_solve_enqueue", "h_solve_wait", "h_post", "h_reject", "h_slide"):
    ms, cnt = pipe.ctx.timing_get(name)
    if cnt:
        print("  %-18s %8.1f us  (n=%d)" % (name, 1e3 * ms / cnt, cnt))
pipe.ctx.close()
