"""debug: phase timestamps of the last be_solve_kernel launch (library built with -DBE_SOLVE_TS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from dynamic_vins_amd import _abi
from dynamic_vins_amd.frontend import Context
from dynamic_vins_amd import backend
import ba_gen, oracle_py
o = oracle_py.load()
ctx = Context(width=640, height=480)
if len(sys.argv) > 1 and sys.argv[1] == '2':
    assert ctx.lib.dv_debug_set(ctx.h, b'two_level', 1) == 0
P = ba_gen.make_window(o, seed=3, nlm=300, max_iters=1, with_prior=True)
out = backend.ba_solve(ctx, P)
print('iters', out.iterations, 'nstate?')
lib = _abi.load()
ts = (C.c_longlong * 32)()
lib.dv_debug_solve_ts.argtypes = [C.POINTER(C.c_longlong)]
print("rc", lib.dv_debug_solve_ts(ts))
t = np.array(ts[:32], dtype=np.int64)
# stamps (be_solve.hip TS(k)): 0 start, 1 scale/grad/tolerance done, 3 landmark diag done, 4 LDL^T blocks loaded, 5 factorised,
# 6 factor stored, 7 back substitution done, 8 Gauss-Newton step complete, 9 dogleg coefficients + delta, 11 H*delta, 12 w.delta,
# 13 candidate written, 10 end
seq = [(0, 1, "scale/grad/tol"), (1, 3, "landmark diag"), (3, 4, "ldlt load"), (4, 5, "ldlt loop"), (5, 6, "ldlt store"), (6, 7, "back-sub"),
       (7, 8, "gn landmarks"), (8, 9, "dogleg (+lazy Cauchy)"), (9, 11, "gemv H*delta"), (11, 12, "w . delta"), (12, 13, "candidate"), (13, 10, "final sums")]
for a, b, name in seq:
    print(f"{name:24s} {(t[b] - t[a]) / 100.0:8.2f} us")
print("total", (t[10] - t[0]) / 100.0)
if t[20]:
    for a, b, name in [(3, 20, "two-level: gather"), (20, 21, "two-level: speed-bias sweep"), (21, 4, "two-level: pose blocks to registers"), (24, 7, "two-level: backward sweep + scatter")]:
        print(f"{name:32s} {(t[b] - t[a]) / 100.0:8.2f} us")
print("ldlt (c)+(a) us", ts[16] / 100.0, " (b) us", ts[17] / 100.0)

if t[31] and t[5] != t[4]:
    print("shader clock during the LDL^T loop: %.0f MHz" % ((t[31] - t[30]) / ((t[5] - t[4]) / 100.0)))

