"""debug: be_solve phase stamps of the LAST solve launch of every frame in the bench workload (library built with -DBE_SOLVE_TS, DVINS_HIP_LIB=...)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim, _abi
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
N = 60
seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
pipe = Pipeline(seq)
lib = _abi.load()
lib.dv_debug_solve_ts.argtypes = [C.POINTER(C.c_longlong)]
seqp = [(15, 0, "accept decision (prologue)"), (0, 3, "scale + gradient"), (3, 4, "ldlt load"), (3, 21, "  mf16: to slots"), (21, 22, "  mf16: tile finish"), (22, 4, "  mf16: barrier"), (4, 5, "ldlt loop"), (5, 6, "ldlt store(+cost/tol)"), (6, 7, "back-sub"),
        (7, 8, "gn landmarks"), (8, 9, "dogleg (+lazy Cauchy)"), (9, 11, "gemv H*delta"), (11, 12, "w . delta"), (12, 13, "candidate"), (13, 10, "final sums"), (15, 10, "total"), (9, 25, "w15: to candidate start"), (25, 26, "w15: pose"), (26, 27, "w15: sb"), (27, 28, "w15: ex"), (28, 29, "w15: rest"), (13, 29, "tid0 ready -> w15 ready"), (29, 10, "w15 ready -> end")]
rows = []
extra = []
for k in range(N):
    pipe.step()
    torch.cuda.synchronize()
    ts = (C.c_longlong * 48)()
    lib.dv_debug_solve_ts(ts)
    t = np.array(ts[:48], dtype=np.int64)
    if k >= 20: rows.append([(t[b] - t[a]) / 100.0 for a, b, _ in seqp])
    if k >= 20: extra.append([t[16] / 100.0, t[17] / 100.0, t[18] / 100.0, t[19] / 100.0, t[20] / 100.0, t[32] / 100.0, t[33] / 100.0, t[34] / 100.0, t[35] / 100.0])
r = np.array(rows)
for i, (_, _, name) in enumerate(seqp):
    print(f"{name:26s} mean {r[:, i].mean():7.2f}  min {r[:, i].min():7.2f}  max {r[:, i].max():7.2f}")
e = np.array(extra)
for i, name in enumerate(["mf16: load + diag tile 0", "mf16: panels (sum over steps)", "mf16: updates + next diag (sum)", "mf16: diag tiles 1.. (owner wave, sum)", "mf16: owner's update before its diag (sum)", "  chain: barrier A -> dk arrived (sum)", "  chain: panel of (k+1, k) (sum)", "  chain: own diagonal update, result landed (sum)", "  chain: barrier B (sum)"]):
    print(f"{name:44s} mean {e[:, i].mean():7.2f}")
