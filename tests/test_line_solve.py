"""The line-only refinement: ceres::Solve of Estimator::OptimizationWithOnlyLine (estimator/estimator.cpp:345-395)
on the problem AddLineResidualBlock builds (:222-253).
CPU (-m "not gpu"): the oracle's problem assembly against an independent numpy restatement of the Cauchy-robustified cost (Plücker
transforms of tests/line_geometry_np.py), the reference's shipped behaviour (zero sqrt_info: returns at once) and invariants.
GPU (-m gpu): dv_line_solve (generic block-diagonal solver, bd_solve.h + be_linesolve.hip) against the oracle: same iteration /
acceptance sequence, costs to 1e-8 relative, parameters: median 1e-10, maximum 1e-6 (conditioning of the representation, see the test)."""
import numpy as np
import pytest

from tests import obj_gen as G

SCENES = [dict(seed=1, max_iters=6), dict(seed=2, orth_noise=0.2, max_iters=8), dict(seed=3, sqrt_info=(0, 0, 0, 0)), dict(seed=4, n_lines=200, max_iters=6),
          dict(seed=5, pix_sigma=0.02, orth_noise=0.1, max_iters=8), dict(seed=6, n_lines=3, empty_lines=0, max_iters=8),
          dict(seed=7, n_lines=700, max_iters=5), dict(seed=8, sqrt_info=(300.0, 20.0, -10.0, 280.0), max_iters=8)]
LONG = [dict(seed=1, max_iters=10), dict(seed=2, orth_noise=0.2, max_iters=30), dict(seed=6, n_lines=3, empty_lines=0, max_iters=25)]


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory):
    return gpu_ctx_factory(width=64, height=48)


def _cost(prob):
    from tests import line_geometry_np as LG
    si = prob.sqrt_info.reshape(2, 2)
    Rbc, tbc = G._qR(prob.ex_pose[3:]), prob.ex_pose[:3]
    c = 0.0
    for o in prob.obs:
        k, f = int(o["line"]), int(o["frame"])
        Rwb, twb = G._qR(prob.pose[f, 3:]), prob.pose[f, :3]
        Rwc, twc = Rwb @ Rbc, Rwb @ tbc + twb
        lc = LG.plk_from_pose(LG.orth_to_plk(prob.orth[k]), Rwc, twc)
        n = lc[:3]
        d = np.array([o["obs"][0] * n[0] + o["obs"][1] * n[1] + n[2], o["obs"][2] * n[0] + o["obs"][3] * n[1] + n[2]]) / np.hypot(n[0], n[1])
        r = si @ d
        c += 0.5 * np.log1p(r @ r)
    return c


def test_oracle_cost_matches_independent_restatement(oracle):
    for kw in SCENES[:5] + LONG[:1]:
        p = G.make_line_scene(**kw)
        c0 = _cost(p)
        s = G.o_line_solve(oracle.lib, p)
        assert abs(s.initial_cost - c0) <= 1e-9 * max(1.0, c0)
        assert abs(s.final_cost - _cost(p)) <= 1e-9 * max(1.0, c0)
        assert s.final_cost <= s.initial_cost


def test_oracle_reference_as_shipped_returns_at_once(oracle):
    """lineProjectionFactor::sqrt_info is never assigned in the reference: zero residuals, zero gradient, ceres stops before
    the first iteration and the line parameters come back untouched"""
    p = G.make_line_scene(seed=3, sqrt_info=(0, 0, 0, 0))
    o0 = p.orth.copy()
    s = G.o_line_solve(oracle.lib, p)
    assert (s.iterations, s.successful, s.termination) == (0, 0, 1) and s.initial_cost == 0.0
    assert np.array_equal(p.orth, o0)


def test_oracle_unobserved_lines_and_poses_never_move(oracle):
    p = G.make_line_scene(seed=2, orth_noise=0.2, max_iters=30)
    o0, pose0 = p.orth.copy(), p.pose.copy()
    s = G.o_line_solve(oracle.lib, p)
    assert s.successful >= 3 and s.final_cost < 0.8 * s.initial_cost
    assert np.array_equal(p.orth[-2:], o0[-2:]) and np.array_equal(p.pose, pose0)
    assert (np.abs(p.orth[:-2] - o0[:-2]).max(axis=1) > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", SCENES, ids=lambda k: "seed%d" % k["seed"])
def test_line_solve_matches_oracle(ctx, oracle, kw):
    from dynamic_vins_amd.backend import line_solve
    ref = G.make_line_scene(**kw)
    dev = ref.clone()
    s_ref = G.o_line_solve(oracle.lib, ref)
    s_dev = line_solve(ctx, dev)
    assert (s_dev.iterations, s_dev.successful, s_dev.termination) == (s_ref.iterations, s_ref.successful, s_ref.termination)
    assert abs(s_dev.initial_cost - s_ref.initial_cost) <= 1e-9 * max(1.0, s_ref.initial_cost)
    assert abs(s_dev.final_cost - s_ref.final_cost) <= 1e-8 * max(1.0, s_ref.initial_cost)
    # the orthonormal representation is badly conditioned near its Euler / asin singularities: the 1e-16 differences between the two
    # implementations grow by one to two orders of magnitude per accepted step on the lines that sit there (1e-11 after the first step on
    # every scene), hence a tight median and a looser maximum
    d = np.abs(dev.orth - ref.orth)
    assert np.median(d) <= 1e-10 and d.max() <= 1e-6


@pytest.mark.gpu
def test_line_solve_first_step_is_tight(ctx, oracle):
    from dynamic_vins_amd.backend import line_solve
    for kw in (dict(seed=1), dict(seed=4, n_lines=200), dict(seed=8, sqrt_info=(300.0, 20.0, -10.0, 280.0))):
        ref = G.make_line_scene(**dict(kw, max_iters=1))
        dev = ref.clone()
        s_ref, s_dev = G.o_line_solve(oracle.lib, ref), line_solve(ctx, dev)
        assert s_ref.successful == s_dev.successful == 1
        assert np.abs(dev.orth - ref.orth).max() <= 1e-9 and abs(s_dev.final_cost - s_ref.final_cost) <= 1e-10 * s_ref.initial_cost


@pytest.mark.gpu
@pytest.mark.parametrize("kw", LONG, ids=lambda k: "seed%d" % k["seed"])
def test_line_solve_long_runs_reach_the_same_cost(ctx, oracle, kw):
    """tens of iterations: the accept / reject sequences may part ways on a knife-edge decision; both must end at the same cost"""
    from dynamic_vins_amd.backend import line_solve
    ref = G.make_line_scene(**kw)
    dev = ref.clone()
    s_ref, s_dev = G.o_line_solve(oracle.lib, ref), line_solve(ctx, dev)
    assert abs(s_dev.final_cost - s_ref.final_cost) <= 1e-5 * s_ref.initial_cost and s_dev.final_cost < 0.7 * s_dev.initial_cost


@pytest.mark.gpu
def test_line_solve_two_view_lines_stay_close(ctx, oracle):
    """lines seen in only two frames (the reference never optimises them: kLineMinObs) have a nearly singular 4x4 block; rounding
    differences of 1e-16 between the two implementations grow by its condition number per accepted step, so only the cost is compared"""
    from dynamic_vins_amd.backend import line_solve
    ref = G.make_line_scene(seed=7, n_lines=300, max_iters=6, min_obs=2)
    dev = ref.clone()
    s_ref, s_dev = G.o_line_solve(oracle.lib, ref), line_solve(ctx, dev)
    assert s_dev.successful >= 1 and abs(s_dev.final_cost - s_ref.final_cost) <= 1e-6 * s_ref.initial_cost
    assert np.median(np.abs(dev.orth - ref.orth)) <= 1e-10


@pytest.mark.gpu
def test_line_solve_factor_order_and_errors(ctx):
    from dynamic_vins_amd.backend import DvinsError, line_solve
    a = G.make_line_scene(seed=5, pix_sigma=0.02, orth_noise=0.1, max_iters=6)
    b = a.clone()
    b.obs = b.obs[np.random.default_rng(0).permutation(len(b.obs))].copy()
    sa, sb = line_solve(ctx, a), line_solve(ctx, b)
    assert (sa.iterations, sa.successful, sa.termination) == (sb.iterations, sb.successful, sb.termination)
    assert np.median(np.abs(a.orth - b.orth)) <= 1e-11 and np.abs(a.orth - b.orth).max() <= 1e-6
    c = G.make_line_scene(seed=5, pix_sigma=0.02, orth_noise=0.1, max_iters=6)
    line_solve(ctx, c)
    assert np.array_equal(c.orth, a.orth)                  # bit-reproducible run to run
    q = a.clone()
    q.obs["frame"][0] = 11
    with pytest.raises(DvinsError, match="out of range"):
        line_solve(ctx, q)
