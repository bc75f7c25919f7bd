"""debug: wall time of dv_ba_solve with constant and with free extrinsic / td blocks on the same window (host-timed, 20 solves each)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamic_vins_amd.backend import ba_solve
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py
ora = oracle_py.load()
ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
kw = dict(seed=31, nlm=260, with_prior=True, feat_vel=True, td_true=0.02, ex_noise=(0.01, 0.005), prior_ex_scale=1.0, max_iters=10)
for fb in (0, 1, 2, 3):
    base = ba_gen.make_window(ora, free_blocks=fb, **kw)
    ba_solve(ctx, base.clone())
    ts, its = [], 0
    for _ in range(20):
        p = base.clone()
        t0 = time.perf_counter(); s = ba_solve(ctx, p); ts.append(time.perf_counter() - t0); its = s.iterations
    ts.sort()
    print("free_blocks %d: %d iterations, median %.0f us per solve, %.0f us per iteration" % (fb, its, 1e6 * ts[10], 1e6 * ts[10] / max(its, 1)))
ctx.close()
