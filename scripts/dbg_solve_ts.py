"""debug: phase timestamps of the last be_solve_kernel launch (library built with -DBE_SOLVE_TS)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from dynamic_vins_amd import _abi
from dynamic_vins_amd.frontend import Context
from dynamic_vins_amd import backend
import ba_gen, oracle_py
o = oracle_py.load()
ctx = Context(width=640, height=480)
P = ba_gen.make_window(o, seed=3, nlm=300, max_iters=1, with_prior=True)
out = backend.ba_solve(ctx, P)
print('iters', out.iterations, 'nstate?')
lib = _abi.load()
ts = (C.c_longlong * 32)()
lib.dv_debug_solve_ts.argtypes = [C.POINTER(C.c_longlong)]
print("rc", lib.dv_debug_solve_ts(ts))
t = np.array(ts[:11], dtype=np.int64)
names = ["scale/grad", "gemv1", "cauchy-lm", "ldlt-load", "ldlt-loop", "ldlt-store", "backsub", "gn-lm", "dogleg-p", "gemv2+cand", "end"]
for k in range(10):
    print(f"{names[k]:12s} {(t[k+1]-t[k]) / 100.0:8.2f} us")
print("total", (t[10] - t[0]) / 100.0)
print("tail: gemv2 %.2f wdot %.2f cand %.2f sums %.2f" % ((ts[11]-ts[9])/100., (ts[12]-ts[11])/100., (ts[13]-ts[12])/100., (ts[10]-ts[13])/100.))
print("ldlt (c)+(a) us", ts[16] / 100.0, " (b) us", ts[17] / 100.0)
