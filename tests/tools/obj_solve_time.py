"""Latency of the per-frame object solve (dv_obj_solve, one persistent workgroup) against the CPU oracle on the same scenes.
Usage: python tests/tools/obj_solve_time.py            (needs the GPU; prints one line per scene)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamic_vins_amd.backend import obj_solve          # noqa: E402
from dynamic_vins_amd.frontend import Context           # noqa: E402
from tests import obj_gen as G, oracle_py               # noqa: E402

ctx = Context(width=64, height=48)
lib = oracle_py.load().lib
ctx.timing_enable(1)
for name, kw in [("typical frame: 10 objects x 100 points, 10 iterations (all rejected)", dict(seed=1, n_obj=10, pts_per_obj=100)),
                 ("detections only: 10 objects, 10 accepted iterations", dict(seed=2, n_obj=10, pts_per_obj=0)),
                 ("mixed: 10 objects x 100 points, 30 iterations", dict(seed=4, n_obj=10, pts_per_obj=100, pose_noise=(0.05, 0.01), max_iters=30)),
                 ("large: 70 objects x 100 points, 15 iterations", dict(seed=10, n_obj=70, pts_per_obj=100, pose_noise=(0.05, 0.01), max_iters=15))]:
    base = G.make_obj_scene(**kw)
    obj_solve(ctx, base.clone())
    ctx.timing_reset()
    reps, t0 = 20, time.perf_counter()
    for _ in range(reps):
        s = obj_solve(ctx, base.clone())
    wall = (time.perf_counter() - t0) / reps * 1e3
    ms, cnt = ctx.timing_get("obj_solve")
    t0 = time.perf_counter()
    for _ in range(5):
        so = G.o_obj_solve(lib, base.clone())
    cpu = (time.perf_counter() - t0) / 5 * 1e3
    print("%-70s points %5d boxes %3d | iterations %2d accepted %2d | kernel %.3f ms  call %.3f ms | oracle (1 core) %.3f ms" % (
        name, len(base.points), len(base.boxes), s.iterations, s.successful, ms / max(cnt, 1), wall, cpu))
ctx.close()
