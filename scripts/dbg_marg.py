import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import oracle_py, ba_gen
from dynamic_vins_amd.frontend import Context
from dynamic_vins_amd.backend import marginalize
o = oracle_py.load()
ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
for kw, mode in [(dict(seed=23, with_prior=True), 1), (dict(seed=21, with_prior=True), 0), (dict(seed=22), 0)]:
    full = ba_gen.make_window(o, **kw)
    ba_gen.oracle_solve(o, full)
    sub = ba_gen.marg_subproblem(full, mode)
    po, Ao, bo = ba_gen.oracle_marginalize(o, sub, mode)
    pd, Ad, bd, diag = marginalize(ctx, sub, mode)
    bo_b, bd_b = ba_gen.prior_to_dict(po, Ao, bo), ba_gen.prior_to_dict(pd, Ad, bd)
    Ao_p, bo_p = ba_gen.permute_prior(bo_b, Ao, bo, bd_b)
    print(kw, mode, "n", pd.n, po.n, "c0", pd.c0, po.c0, "diag", diag)
    for k1, (o1, s1, _) in sorted(bd_b.items()):
        row = []
        for k2, (o2, s2, _) in sorted(bd_b.items()):
            d = np.abs(Ad[o1:o1+s1, o2:o2+s2] - Ao_p[o1:o1+s1, o2:o2+s2]).max()
            row.append("%.0e" % d if d > 1e-3 else ".")
        print(k1, " ".join(row), " b:%.1e" % np.abs(bd[o1:o1+s1]-bo_p[o1:o1+s1]).max())
print("device td row", Ad[-1, :8], Ad[-1, -14:])
print("oracle td row", Ao_p[-1, :8], Ao_p[-1, -14:])
print("device ex1 row", Ad[-2, :8])
print("device b", bd[-3:], "oracle b", bo_p[-3:])
