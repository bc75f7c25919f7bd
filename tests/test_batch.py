"""dv_batch (`-m gpu`): several estimators on one GPU whose window solves share every launch must produce EXACTLY what each estimator produces alone —
the batched kernels run the same code on the same data (argument table in HBM instead of kernel arguments, window index in the grid)."""
import numpy as np
import pytest

from dynamic_vins_amd import sim

pytestmark = pytest.mark.gpu
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)


def make_inputs(seed, phase, use_imu):
    traj = sim.Trajectory()
    fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(3000, seed=7 + seed), max_cnt=150 - 20 * seed, pix_sigma=0.3, seed=3 + seed)
    T0 = 1.0 + phase
    return traj, fs, T0


@pytest.mark.parametrize("use_imu", [1, 0])
def test_batched_estimators_equal_single_estimators(gpu_ctx_factory, use_imu):
    from dynamic_vins_amd.backend import Batch, Estimator
    S, frames, dtf = 3, 32, 0.1
    kw = dict(use_imu=use_imu, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
    single, batched, inputs = [], [], []
    for i in range(S):
        single.append(Estimator(gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5), **kw))
        batched.append(Estimator(gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5), **kw))
        inputs.append(make_inputs(i, 0.37 * i, use_imu))
    batch = Batch([e.ctx for e in batched])
    imu = [sim.imu_stream(tr, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE) for tr, _, T0 in inputs]
    k = [0] * S
    for f in range(frames):
        rows = []
        for i, (tr, fs, T0) in enumerate(inputs):
            t = T0 + f * dtf
            ts, acc, gyr = imu[i]
            while k[i] < len(ts) and ts[k[i]] <= t + 0.011:
                single[i].InputIMU(ts[k[i]], acc[k[i]], gyr[k[i]]); batched[i].InputIMU(ts[k[i]], acc[k[i]], gyr[k[i]]); k[i] += 1
            rows.append((fs.frame(t), t))
        if f == 20:        # one member sits a round out: the others are still solved together
            active = [0, 2]
        else:
            active = list(range(S))
        ref = {i: single[i].ProcessMeasurements(*rows[i])[1] for i in active}
        ref = {i: (s.frame, s.nonlinear, s.iterations, s.initial_cost, s.final_cost) for i, s in ref.items()}
        for i in active:
            assert batched[i].ProcessMeasurementsBegin(*rows[i]) == 0
        batch.enqueue()
        for i in active:
            sb = batched[i].ProcessMeasurementsEnd()
            assert (sb.frame, sb.nonlinear, sb.iterations, sb.initial_cost, sb.final_cost) == ref[i], f"frame {f}, member {i}"
            assert np.array_equal(batched[i].window(), single[i].window()), f"frame {f}, member {i}"
    info = batch.info()
    assert info["batched_rounds"] >= frames - 14, info        # every steady-state round went through the shared launches
    batch.close()
