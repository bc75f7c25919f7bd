"""dv_batch (`-m gpu`): several estimators on one GPU whose window solves share every launch must produce EXACTLY what each estimator produces alone —
the batched kernels run the same code on the same data (argument table in HBM instead of kernel arguments, window index in the grid)."""
import numpy as np
import pytest

from dynamic_vins_amd import sim

pytestmark = pytest.mark.gpu
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)


def make_inputs(seed, phase, use_imu):
    traj = sim.Trajectory()
    fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(3000, seed=7 + seed), max_cnt=max(70, 150 - 20 * seed) + (seed % 3), pix_sigma=0.3, seed=3 + seed)
    T0 = 1.0 + phase
    return traj, fs, T0


@pytest.mark.parametrize("use_imu,S,frames", [(1, 3, 32), (0, 3, 32), (1, 16, 22)])
def test_batched_estimators_equal_single_estimators(gpu_ctx_factory, use_imu, S, frames):
    from dynamic_vins_amd.backend import Batch, Estimator
    dtf = 0.1
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
            active = [i for i in range(S) if i != 1]
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


def test_member_destroyed_before_its_batch_and_abort(gpu_ctx_factory):
    """dv_destroy of a member must detach it from its dv_batch (the batch kept a raw pointer and wrote into freed memory on dv_batch_destroy);
    dv_batch_abort releases threads waiting in dv_batch_arrive instead of letting them block for ever."""
    import threading
    from dynamic_vins_amd.backend import Batch, Estimator
    from dynamic_vins_amd.frontend import Context
    kw = dict(use_imu=0, stereo=1, max_iters=4, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
    ctxs = [Context(width=64, height=64, max_cnt=10, min_dist=5) for _ in range(3)]
    ests = [Estimator(c, **kw) for c in ctxs]
    batch = Batch(ctxs)
    ctxs[1].close()                                   # a member goes first
    batch.enqueue()                                   # nothing pending: must not touch the destroyed member
    # the remaining two members still work through the batch
    traj = sim.Trajectory()
    fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(2000), max_cnt=100, pix_sigma=0.3, seed=3)
    for f in range(14):
        rows, t = fs.frame(1.0 + 0.1 * f), 1.0 + 0.1 * f
        for e in (ests[0], ests[2]):
            assert e.ProcessMeasurementsBegin(rows, t) == 0
        batch.enqueue()
        a, b = ests[0].ProcessMeasurementsEnd(), ests[2].ProcessMeasurementsEnd()
        assert (a.frame, a.iterations, a.final_cost) == (b.frame, b.iterations, b.final_cost)
    # abort: one thread waits in arrive(), the "failed" one aborts instead of arriving
    res = {}

    def waiter():
        try:
            batch.arrive(); res["w"] = "returned"
        except Exception as ex:      # DvinsError
            res["w"] = "raised"
    th = threading.Thread(target=waiter); th.start()
    import time
    time.sleep(0.2)
    batch.abort()
    th.join(timeout=10)
    assert not th.is_alive() and res.get("w") == "raised", res
    batch.close()                                     # after a member is gone and after an abort: no crash
    ctxs[0].close(); ctxs[2].close()


@pytest.mark.parametrize("w,h,S,host", [(752, 480, 5, False), (1242, 375, 21, False), (320, 240, 4, True)])
def test_front_ends_in_shared_launches_equal_single_trackers(gpu_ctx_factory, w, h, S, host):
    """dv_batch_track_enqueue: FeatureTracker::TrackImage of S sequences as ONE launch per stage (job tables in HBM, member index in the grid) must leave every
    member EXACTLY the rows its own dv_track_stereo produces — ids, track counts, undistorted points, velocities, stereo matches, bit for bit, frame after frame;
    21 KITTI-size members is BASELINE.json's config 4 ("seq 0000-0020 batched").  Member 1 sits one round out; host-memory frames take the upload path; a member
    with a mask falls back to its own launches inside the same call."""
    import torch
    from dynamic_vins_amd.backend import Batch
    from dynamic_vins_amd.frontend import DV_MODE_NAIVE, make_cam
    from dynamic_vins_amd import synth
    frames = 9
    cam = make_cam(*sim.cam_tuple(sim.scaled_cam(sim.ZED, w, h, 1280, 720)))
    kw = dict(width=w, height=h, max_cnt=120, min_dist=18, cam0=cam, cam1=cam)
    single = [gpu_ctx_factory(**kw) for _ in range(S)]
    batched = [gpu_ctx_factory(**kw) for _ in range(S)]
    seqs = [synth.PlaneSequence(w, h, seed=11 + i, disparity=4.0 + 0.5 * (i % 5)) for i in range(S)]
    batch = Batch(batched)
    mask = np.full((h, w), 255, np.uint8); mask[:, : w // 3] = 0
    keep = []
    for f in range(frames):
        t = 0.05 * f
        jobs, want = [], {}
        for i in range(S):
            if i == 1 and f == 4:
                continue                                    # a member without a frame this round
            left, right = seqs[i].frame(f)
            masked = (i == 2 and f >= 3)                    # from frame 3 on member 2 tracks in naive mode with a mask: not batchable, same call
            want[i] = single[i].track_stereo(left, right, t, mask if masked else None, DV_MODE_NAIVE if masked else 0)
            if host:
                jobs.append(dict(member=i, gray0=left, gray1=right, t=t, mask=mask if masked else None, mode=DV_MODE_NAIVE if masked else 0))
            else:
                l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
                keep += [l, r]
                jobs.append(dict(member=i, gray0=l.data_ptr(), gray1=r.data_ptr(), t=t, mask=mask if masked else None, mode=DV_MODE_NAIVE if masked else 0))
        torch.cuda.synchronize()
        batch.track_enqueue(jobs)
        for i in want:
            got = batched[i].track_stereo_collect()
            assert len(got) == len(want[i]) and len(got) > 20, f"frame {f}, member {i}: {len(got)} vs {len(want[i])} rows"
            assert got.tobytes() == want[i].tobytes(), f"frame {f}, member {i}"
    info = batch.track_info()
    assert info["rounds"] == frames and info["members_batched"] == S * frames - 1 - (frames - 3) and info["members_single"] == frames - 3, info
    batch.close()
