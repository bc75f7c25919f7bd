"""debug: phase timestamps of the marginalization kernels (library built with -DBE_MARG_TS) on a full window (11 frames, prior, ~300 landmarks)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ba_gen                                               # noqa: E402
import oracle_py                                            # noqa: E402
from dynamic_vins_amd import _abi                           # noqa: E402
from dynamic_vins_amd.backend import marginalize            # noqa: E402
from dynamic_vins_amd.frontend import Context               # noqa: E402

o = oracle_py.load()
ctx = Context(width=64, height=48)
full = ba_gen.make_window(o, seed=25, with_prior=True, nlm=600)
ba_gen.oracle_solve(o, full)
sub = ba_gen.marg_subproblem(full, 0)
for _ in range(3):
    pd, A, b, diag = marginalize(ctx, sub, 0)
print("landmarks in the marginalization", len(sub.landmarks), "n", pd.n)
lib = _abi.load()
ts = (C.c_longlong * 32)()
lib.dv_debug_marg_ts.argtypes = [C.POINTER(C.c_longlong)]
assert lib.dv_debug_marg_ts(ts) == 0
t = list(ts)
for a, b_, name in [(0, 1, "lm: tables + frame geometry"), (1, 2, "lm: residual blocks"), (2, 3, "lm: matrix-core sums"), (3, 4, "lm: w rows (pose columns)"), (4, 5, "lm: per-frame blocks"),
                    (6, 7, "sum: operands + rank term"), (7, 18, "sum: exchange + store"), (8, 9, "finish: load rank term + structured sums"), (9, 10, "finish: prior"), (10, 11, "finish: IMU"), (11, 13, "finish: panel LDL^T of the dropped block"),
                    (13, 14, "finish: copy A', b'"), (14, 15, "finish: write"), (15, 16, "finish: c0 LDL^T")]:
    print("%-32s %8.2f us" % (name, (t[b_] - t[a]) / 100.0))
ctx.close()
