"""debug: phase stamps of be_eval_kernel<true> (landmark block 0, first IMU block, prior block) and be_reduce_kernel (pair block 0, one dense
block) on a full window; library built with -DBE_EVAL_TS -DBE_RED_TS"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ba_gen                                               # noqa: E402
import oracle_py                                            # noqa: E402
from dynamic_vins_amd import _abi, backend                  # noqa: E402
from dynamic_vins_amd.frontend import Context               # noqa: E402

o = oracle_py.load()
ctx = Context(width=64, height=48)
P = ba_gen.make_window(o, seed=3, nlm=263, max_iters=1, with_prior=True)
backend.ba_solve(ctx, P)
print("prior n", P.prior.n if P.prior is not None else 0, "landmarks", len(P.landmarks), "factors", len(P.factors))
lib = _abi.load()
for nm, pairs in [("dv_debug_red_ts", [(0, 1, "reduce pair: landmark loop"), (1, 2, "wave sums + wait for wave 6"), (2, 3, "store"), (0, 16, "pair: start -> wave 6 starts"), (16, 17, "pair: wave 6 index chains"), (8, 9, "reduce dense block")]),
                  ("dv_debug_ev_ts", [(0, 1, "eval lm: geometry"), (1, 2, "eval lm: factors"), (2, 3, "eval lm: packet entries"), (8, 9, "eval imu: stage"), (9, 10, "eval imu: raw"),
                                      (10, 11, "eval imu: whiten"), (11, 12, "eval imu: H, g"), (16, 17, "eval prior: dx"), (17, 18, "eval prior: A dx"), (18, 19, "eval prior: cost, g")])]:
    ts = (C.c_longlong * 64)()
    f = getattr(lib, nm)
    f.argtypes = [C.POINTER(C.c_longlong)]
    assert f(ts) == 0
    for a, b, n in pairs:
        print("%-32s %8.2f us" % (n, (ts[b] - ts[a]) / 100.0))
ctx.close()
