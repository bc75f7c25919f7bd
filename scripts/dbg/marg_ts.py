"""debug: phase stamps of the marginalization kernels (block 0) over the bench workload.  Library built with -DBE_MARG_TS (scripts/dbg/build_ts.sh marg),
DVINS_HIP_LIB=$PWD/dbgbuild/libdvins_hip_ts.so"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim, _abi
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
N = 60
seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
pipe = Pipeline(seq)
lib = _abi.load()
lib.dv_debug_marg_ts.argtypes = [C.POINTER(C.c_longlong)]
seqp = [(0, 1, "lm: tables + geometry"), (1, 2, "lm: residual blocks"), (2, 3, "lm: matrix-core sums (waves 0-4)"), (2, 20, "  setup"), (20, 21, "  loop"), (21, 3, "  stores"), (3, 4, "lm: w rows (pose columns)"), (4, 5, "lm: per-frame blocks"), (0, 5, "lm: total (block 0)"),
        (6, 7, "sum: loads + rank term (mfma)"), (6, 22, "  index setup"), (22, 23, "  loads, adds, products"), (23, 7, "  rest"), (7, 18, "sum: gather + store"), (5, 6, "lm end -> sum start"), (18, 8, "sum end -> finish start"),
        (8, 9, "finish: load sum"), (9, 10, "finish: prior"), (10, 11, "finish: imu"), (11, 13, "finish: eliminate dropped block | MF16: tiles from LDS to registers"), (13, 15, "finish: store A', b' | MF16: factorisation (incl. A', b' out of the registers)"),
        (15, 16, "finish: c0 | MF16: scalars"), (8, 16, "finish: total"), (0, 16, "marginalization: total")]
rows = []
for k in range(N):
    pipe.step()
    torch.cuda.synchronize()
    ts = (C.c_longlong * 32)()
    lib.dv_debug_marg_ts(ts)
    t = np.array(ts[:32], dtype=np.int64)
    if k >= 20: rows.append([(t[b] - t[a]) / 100.0 for a, b, _ in seqp])
r = np.array(rows)
for i, (_, _, name) in enumerate(seqp):
    print(f"{name:36s} mean {r[:, i].mean():7.2f}  min {r[:, i].min():7.2f}  max {r[:, i].max():7.2f}")
