"""debug: iteration counts / durations of the LK waves over the bench workload (library built with -DLK_TS, DVINS_HIP_LIB=...)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim, _abi
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
N = 60
seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
pipe = Pipeline(seq)
lib = _abi.load()
lib.dv_debug_lk_ts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ts = (C.c_ulonglong * 16)()
for k in range(N):
    pipe.step()
    if k == 19:
        torch.cuda.synchronize(); lib.dv_debug_lk_ts(ts, 1)
torch.cuda.synchronize(); lib.dv_debug_lk_ts(ts, 0)
it, lv, _, mx, loop, pre, pts, dur = [ts[i] for i in range(8)]
slow_i, slow_pre, jt, jslow, itmax = ts[8], ts[9], ts[10], ts[11], ts[12]
print("points %d, level passes %d (%.1f per point), iterations %d (%.1f per level pass)" % (pts, lv, lv / pts, it, it / lv))
print("per point: mean %.1f us, max %.1f us; loop %.1f us, before the loop (staging, gradients, A) %.1f us" % (dur / pts / 100.0, mx / 100.0, loop / pts / 100.0, pre / pts / 100.0))
print("I tiles on the border path: %d of %d level passes (pre-loop %.1f us each against %.1f on the fast path); J tiles staged %d (%.2f per level pass), %d on the border path; most iterations in one level pass: %d" % (slow_i, lv, slow_pre / max(slow_i, 1) / 100.0, (pre - slow_pre) / max(lv - slow_i, 1) / 100.0, jt, jt / lv, jslow, itmax))
print("J staging: %.2f us per tile; iterations without it: %.2f us each" % (ts[13] / max(jt, 1) / 100.0, (loop - ts[13]) / max(it, 1) / 100.0))
