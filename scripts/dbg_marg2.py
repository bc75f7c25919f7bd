import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import oracle_py, ba_gen
from dynamic_vins_amd.frontend import Context
from dynamic_vins_amd.backend import marginalize, proj_eval, WindowProblem
o = oracle_py.load()
ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
full = ba_gen.make_window(o, seed=22)
sub = ba_gen.marg_subproblem(full, 0)
F = sub.factors
n = len(F)
out = proj_eval(ctx, F, sub.pose[F["fi"]], sub.pose[F["fj"]], np.tile(sub.ex_pose[0], (n, 1)), np.tile(sub.ex_pose[1], (n, 1)), sub.inv_depth[F["lm"]], np.zeros(n))
print("op max|Jtd|", np.abs(out[:, 52:54]).max(), "max |Jl|", np.abs(out[:, 50:52]).max())
# one landmark, no imu
one = WindowProblem(sub.pose, sub.speed_bias, sub.ex_pose, 0.0, sub.inv_depth, F[sub.landmarks[0]["first"]:sub.landmarks[0]["first"] + sub.landmarks[0]["count"]],
                    np.array([(0, sub.landmarks[0]["count"], 0, sub.landmarks[0]["mask"])], ba_gen.LM_DTYPE), sub.imu[:0], 1, 0, 8, 9.81)
pd, Ad, bd, diag = marginalize(ctx, one, 0)
po, Ao, bo = ba_gen.oracle_marginalize(o, one, 0)
print("one-lm: n", pd.n, po.n, "td row dev", Ad[-1], "\n b", bd[-1], "oracle td row max", np.abs(Ao[-1]).max())
print(diag)
