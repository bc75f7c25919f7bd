import os, sys, time
sys.path.insert(0, '/root/repo')
from dynamic_vins_amd.backend import ba_solve
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py
import torch
ora = oracle_py.load()
ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
base = ba_gen.make_window(ora, seed=31, nlm=200, with_prior=True, max_iters=4)
ts = []
for i in range(12):
    p = base.clone()
    if i == 6: torch.cuda.synchronize()
    t0 = time.perf_counter(); ba_solve(ctx, p); ts.append((time.perf_counter() - t0) * 1e3)
print("ba_solve wall ms:", " ".join("%.2f" % t for t in ts))
ctx.close()
