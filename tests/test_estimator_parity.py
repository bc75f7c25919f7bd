"""GPU end-to-end back-end parity: the product estimator (host bookkeeping + HIP solve/marginalization through
dv_est_*) vs the oracle estimator, fed the same synthetic feature tracks and IMU stream (FeatureSim).
Tolerance: window positions within 1e-5 m and quaternions within 1e-6 on every frame of the run (north_star bar:
1e-3 m ATE); ids / flags / landmark counts / iteration counts identical."""
import numpy as np
import pytest

from dynamic_vins_amd import sim

pytestmark = pytest.mark.gpu

NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)


def run_pair(gpu_ctx_factory, oracle, use_imu, frames, plane=0, max_cnt=150, seed=3, two_phase=False):
    from dynamic_vins_amd.backend import Estimator
    ctx = gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5)
    traj = sim.Trajectory()
    cam = sim.EUROC
    pts = sim.room_points(3000)
    fs = sim.FeatureSim(traj, cam, 752, 480, pts, max_cnt=max_cnt, pix_sigma=0.3, seed=seed)
    kw = dict(use_imu=use_imu, stereo=1, max_iters=8, plane_constraint=plane, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
    ref = oracle.estimator(**kw)
    dev = Estimator(ctx, **kw)
    T0, dtf = 1.0, 0.1
    ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE)
    k = 0
    est_p, ref_p, gt_p = [], [], []
    max_dp = max_dq = 0.0
    for f in range(frames):
        t = T0 + f * dtf
        while k < len(ts) and ts[k] <= t + 0.011:
            ref.input_imu(ts[k], acc[k], gyr[k])
            dev.InputIMU(ts[k], acc[k], gyr[k])
            k += 1
        rows = fs.frame(t)
        rc_o, so = ref.process(rows, t)      # (the oracle consumes IMU samples only up to the frame time, so feeding it early is harmless)
        if two_phase:      # dv_est_process_begin / _end with the NEXT frame's IMU samples arriving while the solve is in flight
            rc_d = dev.ProcessMeasurementsBegin(rows, t)
            while k < len(ts) and ts[k] <= t + dtf + 0.011:
                ref.input_imu(ts[k], acc[k], gyr[k])
                dev.InputIMU(ts[k], acc[k], gyr[k])
                k += 1
            sd = dev.ProcessMeasurementsEnd()
        else:
            rc_d, sd = dev.ProcessMeasurements(rows, t)
        assert rc_o == rc_d == 0
        assert (sd.frame, sd.nonlinear, sd.margin_old, sd.n_landmarks, sd.n_long) == (so.frame, so.nonlinear, so.margin_old, so.n_landmarks, so.n_long), f"frame {f}"
        assert sd.iterations == so.iterations, f"frame {f}: iterations {sd.iterations} vs {so.iterations}"
        Wo, Wd = ref.window(), dev.window()
        max_dp = max(max_dp, np.abs(Wo[:, :3] - Wd[:, :3]).max())
        max_dq = max(max_dq, np.abs(Wo[:, 3:7] - Wd[:, 3:7]).max())
        if sd.nonlinear:
            est_p.append(Wd[10, :3].copy()); ref_p.append(Wo[10, :3].copy()); gt_p.append(traj.p(t))
            # the prior's constant c0 = r0^T r0 carries the 1/lambda-weighted rounding noise of A' (|lambda| ~ 1e-5 on a 1e9 matrix):
            # reproducible only to ~1e-4 relative between two implementations; it shifts the cost, not the minimiser
            assert np.isclose(sd.final_cost, so.final_cost, rtol=2e-3)
    return max_dp, max_dq, np.array(est_p), np.array(ref_p), np.array(gt_p)


@pytest.mark.parametrize("use_imu,frames", [(1, 45), (0, 35)])
def test_estimator_tracks_oracle(gpu_ctx_factory, oracle, use_imu, frames):
    max_dp, max_dq, est, ref, gt = run_pair(gpu_ctx_factory, oracle, use_imu, frames)
    assert max_dp < 1e-5 and max_dq < 1e-6, (max_dp, max_dq)
    ate_dev, _, _ = sim.align_ate(est, gt)
    ate_ref, _, _ = sim.align_ate(ref, gt)
    ate_dev_vs_ref, _, _ = sim.align_ate(est, ref)
    assert ate_dev_vs_ref < 1e-5            # north_star: within 1e-3 m of the reference trajectory
    assert abs(ate_dev - ate_ref) < 1e-5 and ate_dev < 0.05


def test_two_phase_process_matches_oracle(gpu_ctx_factory, oracle):
    max_dp, max_dq, est, ref, gt = run_pair(gpu_ctx_factory, oracle, 1, 30, two_phase=True)
    assert max_dp < 1e-5 and max_dq < 1e-6, (max_dp, max_dq)


def test_spare_slot_path_is_equivalent(oracle):
    """be_solve_fused enqueues max_iters slots and only runs the 3 spare ones (plus gauge fix + marginalization again) when
    the control block says the solve is not finished.  Forced here through the debug switch dv_debug_set(ctx, "short_first_pass", 1):
    the trajectory must still track the oracle, i.e. the intermediate gauge fix / marginalization must not disturb the continued solve."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dynamic_vins_amd.frontend import Context
    made = []

    def factory(**kw):
        c = Context(**kw)
        assert c.lib.dv_debug_set(c.h, b"short_first_pass", 1) == 0
        made.append(c)
        return c
    try:
        dp, dq, est, ref, gt = run_pair(factory, oracle, 1, 30)
    finally:
        for c in made:
            c.close()
    assert dp < 1e-5 and dq < 1e-6, (dp, dq)
