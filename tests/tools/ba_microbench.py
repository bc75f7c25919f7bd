"""BA-only micro-benchmark: one full-window solve (dv_ba_solve: 11 frames, VIO, prior, 10 iterations) for L in {150, 300, 1000} landmarks,
HIP against the CPU oracle's solver on one core.  Lives under tests/tools because the windows come from tests/ba_gen.py (oracle pre-integration).
    python tests/tools/ba_microbench.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ba_gen                                               # noqa: E402
import oracle_py                                            # noqa: E402
from dynamic_vins_amd import backend                        # noqa: E402
from dynamic_vins_amd.frontend import Context               # noqa: E402

o = oracle_py.load()
ctx = Context(width=64, height=48)
out = []
for L in (150, 300, 1000):
    ref = ba_gen.make_window(o, seed=40 + L, nlm=L, max_iters=10, with_prior=True)
    probs = [ref.clone() for _ in range(22)]
    for p in probs[:2]:
        backend.ba_solve(ctx, p)                            # warm-up (code objects, workspace)
    ctx.sync()
    t0 = time.perf_counter()
    its = 0
    for p in probs[2:]:
        its += backend.ba_solve(ctx, p).iterations
    ctx.sync()
    dt = (time.perf_counter() - t0) / 20
    c = ref.clone()
    t1 = time.perf_counter()
    so = ba_gen.oracle_solve(o, c)
    dc = time.perf_counter() - t1
    out.append(dict(landmarks=L, residual_blocks=int(len(ref.factors)), iterations=its / 20, hip_ms_per_solve=round(dt * 1e3, 4), hip_us_per_iteration=round(dt / (its / 20) * 1e6, 1),
                    oracle_ms_per_solve_1core=round(dc * 1e3, 2), oracle_iterations=int(so.iterations)))
    print(out[-1])
ctx.close()
if len(sys.argv) > 1:
    json.dump(dict(what="dv_ba_solve on one full window (n = 165, prior, 10 iterations max), upload + solve + download, one stream; CPU = tests' oracle solver, 1 core, -O2", rows=out), open(sys.argv[1], "w"), indent=1)
