"""Free extrinsic / td blocks in the window solve (estimate_extrinsic 1, estimate_td 1: Estimator::AddBodyParameterBlock, estimator/estimator.cpp:87-100).
No shipped YAML switches them on; the product carries them beside the default path (csrc/be_ext.hip) on the generic factorisation (n up to 178).
Checked three ways: (1) the 13 extra rows / columns of the reduced camera system against finite differences of its own right-hand side on a residual-free window
(independent of the oracle), (2) the untouched part of the system bit for bit against the constant-block evaluation, (3) the whole trust-region solve against the
oracle's (same bars as tests/test_back_parity.py: iteration counts equal, costs 1e-7 relative, states 1e-6 absolute) and the estimator run against the oracle's."""
import numpy as np
import pytest

from tests import ba_gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu_ctx_factory):
    return gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5)


def _qmul(a, b):          # x y z w
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _ex_plus(ex7, d6):      # PoseLocalParameterization::Plus (utils/pose_local_parameterization.cpp): p + dp, q * [1, dtheta / 2]
    out = ex7.copy()
    out[:3] += d6[:3]
    dq = np.array([d6[3] / 2, d6[4] / 2, d6[5] / 2, 1.0])
    q = _qmul(ex7[3:], dq / np.linalg.norm(dq))
    out[3:] = q / np.linalg.norm(q)
    return out


@pytest.mark.parametrize("use_imu", [1, 0])
def test_ext_columns_match_finite_differences_of_the_right_hand_side(ctx, oracle, use_imu):
    """On a window whose reprojection residuals vanish, the reduced right-hand side g(x) = g_p - sum_l w_l g_l / h_l moves with the extrinsics and td exactly as the
    Gauss-Newton reduced system says: dg = S[:, ext] d(ext).  The finite difference only needs dv_ba_eval at constant blocks' values, so it checks J_ex / J_td, the ext
    packets and be_reduce_ext against the arithmetic of the DEFAULT path."""
    from dynamic_vins_amd.backend import ba_eval
    kw = dict(seed=41, nlm=150, use_imu=use_imu, nframes=11 if use_imu else 8, pix_sigma=0.0, pose_noise=(0.0, 0.0), depth_noise=0.0, feat_vel=True, td_true=0.0)
    base = ba_gen.make_window(oracle, free_blocks=3, **kw)
    _, S, g = ba_eval(ctx, base)
    n = S.shape[0]
    n_body = n - 13
    assert n_body == (11 * 15 if use_imu else 7 * 6)
    assert np.allclose(S, S.T, rtol=0, atol=1e-9 * np.abs(S).max())
    h = 1e-6
    for q in range(13):
        gs = []
        for sgn in (1.0, -1.0):
            p = base.clone()
            if q < 12:
                d = np.zeros(6)
                d[q % 6] = sgn * h
                p.ex_pose[q // 6] = _ex_plus(p.ex_pose[q // 6], d)
            else:
                p.td[0] += sgn * h
            gs.append(ba_eval(ctx, p)[2])
        fd = (gs[0] - gs[1]) / (2 * h)
        col = S[:, n_body + q]
        scale = max(np.abs(col).max(), 1.0)
        assert np.allclose(fd, col, rtol=0, atol=2e-5 * scale), (q, np.abs(fd - col).max() / scale)


def test_body_part_of_the_system_is_untouched(ctx, oracle):
    from dynamic_vins_amd.backend import ba_eval
    kw = dict(seed=42, nlm=200, with_prior=True, feat_vel=True, td_true=0.01, ex_noise=(0.01, 0.005), prior_ex_scale=1.0)
    c0, S0, g0 = ba_eval(ctx, ba_gen.make_window(oracle, free_blocks=0, **kw))
    for fb, extra in ((1, 12), (2, 1), (3, 13)):
        c, S, g = ba_eval(ctx, ba_gen.make_window(oracle, free_blocks=fb, **kw))
        assert S.shape[0] == S0.shape[0] + extra
        nb = S0.shape[0]
        assert c == c0 and np.array_equal(S[:nb, :nb], S0) and np.array_equal(g[:nb], g0)
        assert np.all(np.diag(S)[nb:] > 0)


FREE_CASES = [dict(seed=31, free_blocks=3), dict(seed=32, free_blocks=1), dict(seed=33, free_blocks=2), dict(seed=34, free_blocks=3, nlm=300, outlier_ratio=0.05),
              dict(seed=35, free_blocks=3, nlm=1000, max_iters=5), dict(seed=36, free_blocks=1, plane_kind=1)]


@pytest.mark.parametrize("kw", FREE_CASES, ids=[str(i) for i in range(len(FREE_CASES))])
def test_ba_solve_with_free_blocks_matches_oracle(ctx, oracle, kw):
    from dynamic_vins_amd.backend import ba_solve
    kw = dict(dict(with_prior=True, feat_vel=True, td_true=0.02, ex_noise=(0.01, 0.005), prior_ex_scale=1.0, max_iters=10), **kw)
    ref = ba_gen.make_window(oracle, **kw)
    dev = ref.clone()
    ex0, td0 = ref.ex_pose.copy(), ref.td[0]
    so = ba_gen.oracle_solve(oracle, ref)
    sd = ba_solve(ctx, dev)
    assert (sd.iterations, sd.successful, sd.termination) == (so.iterations, so.successful, so.termination)
    assert np.isclose(sd.initial_cost, so.initial_cost, rtol=1e-9)
    assert np.isclose(sd.final_cost, so.final_cost, rtol=1e-7)
    assert so.final_cost < so.initial_cost
    assert np.allclose(dev.pose, ref.pose, rtol=0, atol=1e-6)
    assert np.allclose(dev.speed_bias, ref.speed_bias, rtol=0, atol=1e-6)
    assert np.allclose(dev.inv_depth, ref.inv_depth, rtol=0, atol=1e-6)
    assert np.allclose(dev.ex_pose, ref.ex_pose, rtol=0, atol=1e-6) and np.isclose(dev.td[0], ref.td[0], rtol=0, atol=1e-6)
    fb = kw["free_blocks"]
    assert (np.abs(ref.ex_pose - ex0).max() > 1e-4) == bool(fb & 1) and (abs(ref.td[0] - td0) > 1e-5) == bool(fb & 2)      # the free blocks moved, the constant ones did not
    if not fb & 1:
        assert np.array_equal(dev.ex_pose, ex0)
    if not fb & 2:
        assert dev.td[0] == td0


def test_free_block_solve_is_reproducible_and_leaves_the_default_path_alone(ctx, oracle):
    from dynamic_vins_amd.backend import ba_solve
    kw = dict(seed=37, with_prior=True, feat_vel=True, td_true=0.02, ex_noise=(0.01, 0.005), prior_ex_scale=1.0, max_iters=6)
    a, b = ba_gen.make_window(oracle, free_blocks=3, **kw), ba_gen.make_window(oracle, free_blocks=3, **kw)
    d0 = ba_gen.make_window(oracle, free_blocks=0, **kw)
    ba_solve(ctx, d0)
    ba_solve(ctx, a)
    d1 = ba_gen.make_window(oracle, free_blocks=0, **kw)
    ba_solve(ctx, d1)
    ba_solve(ctx, b)
    assert np.array_equal(a.pose, b.pose) and np.array_equal(a.ex_pose, b.ex_pose) and a.td[0] == b.td[0] and np.array_equal(a.inv_depth, b.inv_depth)
    assert np.array_equal(d0.pose, d1.pose) and np.array_equal(d0.inv_depth, d1.inv_depth)      # a constant-block solve before and after a free-block one: same bits


@pytest.mark.parametrize("estimate", [3, 1, 2])
def test_estimator_with_free_extrinsics_and_td_tracks_oracle(gpu_ctx_factory, oracle, estimate):
    """dv_est_config::estimate against the oracle's estimator with the same switch: the extrinsics handed to both are 8 mm / 0.3 degrees off the simulator's, openExEstimation
    opens at the first full window with |Vs[0]| > 0.2 and stays open (estimator.cpp:87-95), td is free while |Vs[0]| >= 0.2 (:98-100); Double2vector hands the solved blocks
    to the next frame's triangulation, factors (cur_td) and marginalization.  Same bars as tests/test_estimator_parity.py."""
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.backend import Estimator
    from tests.test_estimator_parity import NOISE
    ctx = gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5)
    class Excited(sim.Trajectory):          # the figure-8 of the other tests is planar (yaw + 0.05 rad of roll / pitch): the extrinsic translation along the yaw axis is
        def ypr(self, t):                   # unobservable on it and wanders by metres; with 0.25 / 0.2 rad of pitch / roll it settles within centimetres of the simulator's
            y = sim.Trajectory.ypr(self, t)
            return np.array([y[0], 0.25 * np.sin(2.3 * t), 0.2 * np.sin(3.1 * t + 1.0)])
    traj = Excited()
    fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(3000), max_cnt=150, pix_sigma=0.3, seed=5)
    rng = np.random.default_rng(7)
    ric = [sim.R_IC @ ba_gen.small_rot(rng.normal(0, 0.005, 3)) for _ in range(2)]
    tic = [np.asarray(sim.T_IC0) + rng.normal(0, 0.008, 3), np.asarray(sim.T_IC1) + rng.normal(0, 0.008, 3)]
    kw = dict(use_imu=1, stereo=1, max_iters=8, ric=ric, tic=tic, estimate=estimate, **NOISE)
    ref, dev = oracle.estimator(**kw), Estimator(ctx, **kw)
    frames, T0, dtf = 40, 1.0, 0.1
    ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE)
    k = 0
    max_dp = max_dq = 0.0
    moved_ex = moved_td = False
    for f in range(frames):
        t = T0 + f * dtf
        while k < len(ts) and ts[k] <= t + 0.06:          # (the frame's interval ends at t + td: the samples must reach past it whatever td has become)
            ref.input_imu(ts[k], acc[k], gyr[k])
            dev.InputIMU(ts[k], acc[k], gyr[k])
            k += 1
        rows = fs.frame(t)
        rc_o, so = ref.process(rows, t)
        rc_d, sd = dev.ProcessMeasurements(rows, t)
        assert rc_o == rc_d == 0
        assert (sd.frame, sd.nonlinear, sd.margin_old, sd.n_landmarks, sd.n_long) == (so.frame, so.nonlinear, so.margin_old, so.n_landmarks, so.n_long), f"frame {f}"
        assert sd.iterations == so.iterations, f"frame {f}: iterations {sd.iterations} vs {so.iterations}"
        Wo, Wd = ref.window(), dev.window()
        max_dp = max(max_dp, np.abs(Wo[:, :3] - Wd[:, :3]).max())
        max_dq = max(max_dq, np.abs(Wo[:, 3:7] - Wd[:, 3:7]).max())
        (ro, to, tdo), (rd, tdv, tdd) = ref.extrinsics(), dev.extrinsics()
        assert np.abs(ro - rd).max() < 1e-6 and np.abs(to - tdv).max() < 1e-6 and abs(tdo - tdd) < 1e-6, f"frame {f}"
        moved_ex |= np.abs(to - np.array(tic)).max() > 1e-5
        moved_td |= abs(tdo) > 1e-6
    assert max_dp < 1e-5 and max_dq < 1e-6, (max_dp, max_dq)
    assert moved_ex == bool(estimate & 1) and moved_td == bool(estimate & 2)
    if estimate & 1:
        assert np.abs(to - np.array([sim.T_IC0, sim.T_IC1])).max() < 0.05          # (it stays with the simulator's extrinsics)


def test_free_blocks_inside_a_dv_batch(gpu_ctx_factory):
    """A dv_batch member whose solve frees the extrinsics takes the generic factorisation, so the round it is in falls back to the members' own launches (be_api.hip
    batch_enqueue_impl: mixed kernel variants): every member, the free one and the constant ones beside it, still produces exactly what it produces alone."""
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.backend import Batch, Estimator
    from tests.test_batch import NOISE, make_inputs
    S, frames, dtf = 3, 24, 0.1
    kws = [dict(use_imu=1, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], estimate=3 if i == 0 else 0, **NOISE) for i in range(S)]
    single = [Estimator(gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5), **kws[i]) for i in range(S)]
    batched = [Estimator(gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5), **kws[i]) for i in range(S)]
    inputs = [make_inputs(i, 0.37 * i, 1) for i in range(S)]
    batch = Batch([e.ctx for e in batched])
    imu = [sim.imu_stream(tr, T0 - 0.05, T0 + frames * dtf + 0.2, 200.0, **NOISE) for tr, _, T0 in inputs]
    k = [0] * S
    for f in range(frames):
        rows = []
        for i, (tr, fs, T0) in enumerate(inputs):
            t = T0 + f * dtf
            ts, acc, gyr = imu[i]
            while k[i] < len(ts) and ts[k[i]] <= t + 0.06:
                single[i].InputIMU(ts[k[i]], acc[k[i]], gyr[k[i]]); batched[i].InputIMU(ts[k[i]], acc[k[i]], gyr[k[i]]); k[i] += 1
            rows.append((fs.frame(t), t))
        ref = [single[i].ProcessMeasurements(*rows[i])[1] for i in range(S)]
        ref = [(s.frame, s.nonlinear, s.iterations, s.initial_cost, s.final_cost) for s in ref]
        for i in range(S):
            assert batched[i].ProcessMeasurementsBegin(*rows[i]) == 0
        batch.enqueue()
        for i in range(S):
            sb = batched[i].ProcessMeasurementsEnd()
            assert (sb.frame, sb.nonlinear, sb.iterations, sb.initial_cost, sb.final_cost) == ref[i], f"frame {f}, member {i}"
            assert np.array_equal(batched[i].window(), single[i].window()), f"frame {f}, member {i}"
            for a, b in zip(batched[i].extrinsics(), single[i].extrinsics()):
                assert np.array_equal(a, b)
    assert np.abs(batched[0].extrinsics()[1] - np.array([sim.T_IC0, sim.T_IC1])).max() > 1e-6 and np.array_equal(batched[1].extrinsics()[1], np.array([sim.T_IC0, sim.T_IC1]))
    info = batch.info()
    assert info["single_rounds"] >= 10, info          # (openExEstimation opens with the first full window: every steady-state round of this batch ran on the members' own launches)
    batch.close()


def test_unknown_bits_are_refused(ctx, oracle, gpu_ctx_factory):
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.backend import Estimator, ba_solve
    from dynamic_vins_amd.frontend import DvinsError
    p = ba_gen.make_window(oracle, seed=38, nlm=40, max_iters=2)
    p.c.free_blocks = 4
    with pytest.raises(DvinsError, match="free_blocks"):
        ba_solve(ctx, p)
    p.c.free_blocks = 0
    assert ba_solve(ctx, p).iterations >= 1          # the context is usable afterwards
    with pytest.raises(DvinsError, match="estimate"):
        Estimator(gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5), use_imu=1, stereo=1, max_iters=4, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], estimate=4)
