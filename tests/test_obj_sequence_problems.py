"""dv_obj_solve against dvo_obj_solve on object-solve problems a REAL dynamic sequence produced (tests/golden/obj_sequence_problems.npz, written by
tests/tools/obj_problem_dump.py from the oracle's estimator, CPU only): the escort scene at 1280x720, frames 60 - 131.  The random problems of tests/obj_gen.py are well
posed; these include the regime the long dynamic run found (tests/tools/longrun_parity.py dynamic 600 1280 720): one box classified static for ~25 frames while it travels
with the camera — its enclose factors (a hinge: box_factor.cpp:523-584) sit far outside the box, the solve stagnates.  Same inputs, bit for bit, on both sides: whatever
the two solvers' results differ by here is the solver's own, not the history's."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "obj_sequence_problems.npz")


def problems():
    from dynamic_vins_amd.backend import ObjProblem
    z = np.load(GOLD)
    for k in z["frames"]:
        meta = z[f"f{k}_meta"]
        yield int(k), ObjProblem(z[f"f{k}_state"], z[f"f{k}_dims"], z[f"f{k}_body"], z[f"f{k}_R_bc"], z[f"f{k}_boxes"], z[f"f{k}_points"], max_iters=int(meta[0]), plane_kind=int(meta[1])), \
            z[f"f{k}_static"], z[f"f{k}_summary"]


def test_fixture_is_what_the_oracle_solves_today(oracle):
    """CPU: the stored summaries are the oracle's own (the fixture has not drifted from the solver it was dumped from)"""
    from tests.obj_gen import o_obj_solve
    n = 0
    for k, p, static, summ in problems():
        s = o_obj_solve(oracle.lib, p)
        assert (s.iterations, s.termination) == (int(summ[0]), int(summ[1])), (k, s.iterations, s.termination, summ.tolist())          # (summary4 = iterations, termination, initial cost, final cost)
        assert abs(s.initial_cost - summ[2]) <= 1e-9 * max(1.0, abs(summ[2])) and abs(s.final_cost - summ[3]) <= 1e-9 * max(1.0, abs(summ[3])), (k, s.initial_cost, s.final_cost, summ.tolist())
        n += 1
    assert n >= 8


@pytest.mark.gpu
def test_obj_solve_on_sequence_problems(oracle):
    from dynamic_vins_amd.backend import obj_solve
    from dynamic_vins_amd.frontend import Context
    from tests.obj_gen import o_obj_solve
    ctx = Context(width=64, height=48)
    worst = {}
    try:
        for k, p, static, summ in problems():
            q = p.clone()
            so = o_obj_solve(oracle.lib, p)
            sd = obj_solve(ctx, q)
            dp = float(np.abs(p.state[:, :, :3] - q.state[:, :, :3]).max()); dq = float(np.abs(p.state[:, :, 3:] - q.state[:, :, 3:]).max()); dd = float(np.abs(p.dims - q.dims).max())
            worst[k] = dict(dp=dp, dq=dq, ddims=dd, it=(int(so.iterations), int(sd.iterations)), cost0=(so.initial_cost, sd.initial_cost), cost1=(so.final_cost, sd.final_cost), static=static.tolist())
            # same inputs: initial costs to rounding; final states to the operator bar of tests/test_obj_parity.py
            assert abs(so.initial_cost - sd.initial_cost) <= 1e-9 * max(1.0, abs(so.initial_cost)), (k, worst[k])
        print(worst)
        for k, wv in worst.items():
            assert wv["it"][0] == wv["it"][1] and wv["dp"] < 1e-6 and wv["dq"] < 1e-6 and wv["ddims"] < 1e-6, (k, wv)
    finally:
        ctx.close()
