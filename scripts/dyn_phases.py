"""Host wall-clock per phase of the DYNAMIC pipeline (dv_timing_enable(ctx, -1): host scopes only, no extra synchronisation) + Python-side phases."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamic_vins_amd import sim                                     # noqa: E402
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence    # noqa: E402

W, H, STEPS, WARM = 1280, 720, 100, 14
SCENE = sys.argv[1] if len(sys.argv) > 1 else "escort"          # escort (bench default: 4 boxes in view in every frame) | room (round-2 scene)
seq = DynamicSequence(W, H, sim.ZED, WARM + STEPS + 1, rate=20.0, device="cuda:0", boxes=("escort", 4) if SCENE == "escort" else None)
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
for _ in range(WARM):
    pipe.step()
pipe.ctx.timing_enable(-1)
pipe.ctx.sync(); torch.cuda.synchronize()
ph = dict(collect=0.0, inst_collect=0.0, begin=0.0, enqueue=0.0, end=0.0)
t0 = time.perf_counter()
nobj = 0
for i in range(STEPS):
    k, s = pipe.next, pipe.seq
    a = time.perf_counter()
    pre, pipe._prefetched = pipe._prefetched, None          # (Pipeline.step collects frame k+1's rows while the BA of frame k runs; this loop times the phases in sequence)
    rows = pre[0] if pre is not None else pipe.ctx.track_stereo_collect()
    b = time.perf_counter()
    insts, ifeats, pts = pre[1:] if pre is not None else pipe.ctx.inst_track_collect()
    c = time.perf_counter()
    t = s.times[k]
    pipe._feed_imu(t)
    pipe.est.ProcessMeasurementsDynamicBegin(rows, t, insts, ifeats, pts)
    d = time.perf_counter()
    if k + 1 < len(s.frames):
        pipe._enqueue(k + 1)
        pipe._feed_imu(s.times[k + 1])
    e = time.perf_counter()
    st = pipe.est.ProcessMeasurementsEnd()
    f = time.perf_counter()
    pipe.next += 1
    ph["collect"] += b - a; ph["inst_collect"] += c - b; ph["begin"] += d - c; ph["enqueue"] += e - d; ph["end"] += f - e
    nobj += len(insts)
    if i % 25 == 0:
        print("frame", k, "objects", len(insts), "obj feats", len(ifeats), "extra pts", len(pts))
pipe.ctx.sync()
dt = time.perf_counter() - t0
print("%.1f frames/s, %.1f us per frame, %.2f objects/frame" % (STEPS / dt, dt / STEPS * 1e6, nobj / STEPS))
for k2, v in ph.items():
    print("  py %-14s %8.1f us" % (k2, v / STEPS * 1e6))
for name in ("h_imu", "h_add_features", "h_triangulate", "h_build", "h_solve_begin", "h_solve_upload", "h_solve_enqueue", "h_dynamic", "h_dyn_push", "h_dyn_propagate", "h_dyn_triangulate", "h_dyn_initial", "h_dyn_build", "h_dyn_solve_begin", "h_dynamic_finish", "h_dyn_solve_wait", "h_dyn_reject", "h_dyn_slide", "h_dyn_finish_frame", "h_solve_wait", "h_post", "h_reject", "h_slide",
             "h_process_begin", "h_process_end"):
    ms, cnt = pipe.ctx.timing_get(name)
    if cnt:
        print("  %-18s %8.1f us  (n=%d)" % (name, 1e3 * ms / cnt, cnt))
pipe.ctx.close()
