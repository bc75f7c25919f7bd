"""dv_runner (`-m gpu`): the per-frame host loop in C++ inside the library must leave EXACTLY what the Python pipeline leaves — same calls in the same order on the
same contexts — for one sequence, for several sequences on their own streams, and for dv_batch groups driven by one or two host threads."""
import numpy as np
import pytest

from dynamic_vins_amd import sim

pytestmark = pytest.mark.gpu


def make(n_seq, frames, w=752, h=480):
    from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seqs = [SyntheticSequence(w, h, cam, frames, rate=20.0, phase=1.3 * i) for i in range(n_seq)]
    return seqs, [Pipeline(q, max_cnt=150, min_dist=30, max_iters=8) for q in seqs]


@pytest.mark.parametrize("group_size,threads", [(0, 1), (3, 1), (2, 2), (4, 2), (2, 4)])
def test_runner_equals_python_pipeline(group_size, threads):
    """(4, 2) / (2, 4): more threads than groups are asked for — teams of two host threads per group (the default since round 5; dv_runner_set "teams" 0 = one thread per group)"""
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import Pipeline
    S, frames = 4 if threads >= 2 else 3, 30
    seqs, pipes = make(S, frames)
    ref = [Pipeline(q, max_cnt=150, min_dist=30, max_iters=8) for q in seqs]
    runner = Runner(pipes, group_size=group_size, threads=threads)
    runner.run(frames - 1)
    for i in range(S):
        for _ in range(frames - 1):
            ref[i].step()
        st, poses, iters, fr = runner.get(i)
        assert fr == frames - 1 and st.frame == ref[i].last_state.frame and st.nonlinear == ref[i].last_state.nonlinear
        assert np.array_equal(np.ctypeslib.as_array(st.window), ref[i].est.window()), f"sequence {i}: window states differ"
        want = np.array(ref[i].poses)
        assert len(poses) == len(want) >= frames - 14
        assert np.array_equal(poses[:, 1:], want) and np.array_equal(poses[:, 0], np.array(ref[i].pose_times))
    runner.close()
    for p in pipes + ref:
        p.ctx.close()


def _trajectories(group_size, threads, S, frames, keys=(), teams=False):
    from dynamic_vins_amd.backend import Runner
    _, pipes = make(S, frames)
    for p in pipes:
        for k in keys:
            assert p.ctx.lib.dv_debug_set(p.ctx.h, k.encode(), 1) == 0, k
    r = Runner(pipes, group_size=group_size, threads=threads)
    r.set("teams", 1 if teams else 0)
    r.run(frames - 1)
    out = [r.frames(i) for i in range(S)]
    r.close()
    for p in pipes:
        p.ctx.close()
    return out


def test_two_groups_in_flight_are_bit_identical_to_the_single_thread_run():
    """The configuration class in which round 4 found single members leaving their trajectory (two dv_batch groups in flight, one host thread each): every member's per-frame
    record [t, pose, flag] must equal the unbatched single-thread run's, bit for bit, over 60 frames, in each of four repetitions.  Cause (fixed): be_accept_body let thread 0
    store into the control block before every wave of the workgroup had loaded it — a wave started late on a busy CU saw pending == 0 and skipped the body, wave 0 summed
    unwritten LDS and a quarter of the state was copied.  Before the fix 8 of 30 such runs differed at 1280x720 with the group's shared accept + gauge launch, after it 0 of 60."""
    S, frames = 8, 60
    want = _trajectories(0, 1, S, frames)
    for rep in range(4):
        got = _trajectories(4, 2, S, frames)
        for i in range(S):
            assert got[i].shape == want[i].shape and np.array_equal(got[i], want[i]), f"repetition {rep}: sequence {i} differs from the single-thread run"


def test_members_own_accept_gauge_launches_give_the_same_bits():
    """dv_debug_set 'batch_single_tail': the accept decision + gauge fix + download of a group through every member's own launch of the same two bodies (the A/B switch that
    located the defect above) — same bits as the shared launch and as the single-thread run"""
    S, frames = 8, 60
    want = _trajectories(0, 1, S, frames)
    for rep in range(2):
        got = _trajectories(4, 2, S, frames, keys=("batch_single_tail",))
        for i in range(S):
            assert np.array_equal(got[i], want[i]), f"repetition {rep}: sequence {i} differs from the single-thread run"


def test_team_path_is_bit_identical_to_the_single_thread_run():
    """several host threads per group (dv_runner_set 'teams', opt-in): the failures first blamed on it (round 4) came from the race in be_accept_body (previous test); the
    team path must reproduce the single-thread run bit for bit — two groups of four with teams of two, and one group of eight with a
    team of four, three repetitions each"""
    S, frames = 8, 60
    want = _trajectories(0, 1, S, frames)
    for gs, T in ((4, 4), (8, 4), (4, 4), (8, 4), (4, 4), (8, 4)):
        got = _trajectories(gs, T, S, frames, teams=True)
        for i in range(S):
            assert np.array_equal(got[i], want[i]), f"groups of {gs} on {T} threads: sequence {i} differs from the single-thread run"


def test_batch_stage_timing_leaves_the_results_alone():
    """dv_runner_batch_timing / dv_batch_timing (the batched regime's roofline line): HIP events around one steady-state slot of every round — positive stage
    times, a round count, the windows of a launch; and the run with the events on leaves the same bits as the run without them"""
    from dynamic_vins_amd.backend import Runner
    frames = 26
    seqs, pipes = make(3, frames)
    _, plain = make(3, frames)
    timed = Runner(pipes, group_size=3, threads=1)
    ref = Runner(plain, group_size=3, threads=1)
    timed.run(13); ref.run(13)
    t_ms, rounds, wins = timed.batch_timing(1)
    assert rounds == 0                                   # switched on just now
    timed.run(frames - 14); ref.run(frames - 14)
    t_ms, rounds, wins = timed.batch_timing(1)
    assert rounds >= 5 and wins == 3 and (t_ms > 0).all() and (t_ms < 5.0).all(), (t_ms, rounds, wins)
    for i in range(3):
        a, b = timed.get(i), ref.get(i)
        assert np.array_equal(a[1], b[1]) and np.array_equal(np.ctypeslib.as_array(a[0].window), np.ctypeslib.as_array(b[0].window))
    timed.close(); ref.close()
    for p in pipes + plain:
        p.ctx.close()


def test_runner_shared_front_end_launches_leave_the_same_bits():
    """dv_runner groups with dv_batch_track_enqueue (default) against the same groups with one set of tracking launches per sequence (`batch_front` 0), five KITTI-size
    sequences, frame gate of the non-KITTI datasets on one of the runs' configurations: identical trajectories, window states and row counts"""
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
    w, h, S, frames = 1242, 375, 5, 28
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seqs = [SyntheticSequence(w, h, cam, frames, rate=20.0, phase=0.9 * i) for i in range(S)]
    for stride in (1, 2):
        a = [Pipeline(q, max_cnt=150, min_dist=30, max_iters=8, use_imu=1, ba_stride=stride) for q in seqs]
        b = [Pipeline(q, max_cnt=150, min_dist=30, max_iters=8, use_imu=1, ba_stride=stride) for q in seqs]
        ra, rb = Runner(a, group_size=S, threads=1), Runner(b, group_size=S, threads=1)
        rb.set("batch_front", 0)
        ra.run(frames - 1); rb.run(frames - 1)
        ia, ib = ra.track_info(), rb.track_info()
        assert ia["members_batched"] >= S * (frames - 2) and ia["members_single"] == 0, ia
        assert ib["members_batched"] == 0, ib
        for i in range(S):
            sa, pa, _, fa = ra.get(i); rows_a = ra.last_rows
            sb, pb, _, fb = rb.get(i); rows_b = rb.last_rows
            assert fa == fb and rows_a == rows_b and len(pa) == len(pb) >= (frames - 1) // stride - 11
            assert np.array_equal(pa, pb) and np.array_equal(np.ctypeslib.as_array(sa.window), np.ctypeslib.as_array(sb.window)), (stride, i)
        ra.close(); rb.close()
        for p in a + b:
            p.ctx.close()


def test_pinned_host_frames_read_in_place_give_the_same_bits():
    """DV_MEM_PINNED (dv_seq_input::mem): caller-pinned host frames are read IN PLACE by the pyramid kernel over PCIe — no staging copy, no copy engine.  The runner fed pinned
    frames, pageable-style DV_MEM_HOST frames (hipMemcpy2DAsync) and HBM-resident frames must leave the same rows and states, bit for bit."""
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
    w, h, frames = 752, 480, 26
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seq = SyntheticSequence(w, h, cam, frames, rate=20.0)
    outs = []
    for hf in (False, True, "engine"):
        p = Pipeline(seq, max_cnt=150, min_dist=30, max_iters=8, use_imu=1)
        r = Runner([p], host_frames=hf)
        r.run(frames - 1)
        st, poses, iters, fr = r.get(0)
        outs.append((np.ctypeslib.as_array(st.window).copy(), poses.copy(), r.frames(0).copy(), r.row_log(0).copy() if hasattr(r, "row_log") else None))
        r.close(); p.ctx.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
        if o[3] is not None:
            assert np.array_equal(o[3], outs[0][3])
    assert len(outs[0][1]) >= frames - 13


@pytest.mark.parametrize("tracker_thread,calls,stride", [(1, (43,), 1), (0, (43,), 1), (1, (9, 1, 20, 13), 1), (1, (43,), 2), (0, (20, 23), 2)])
def test_runner_static_feedback_and_frame_stride_equal_the_python_pipeline(tracker_thread, calls, stride):
    """the T3 -> T2 feedback of para::is_static_inst_as_background (dv_seq_dynamic::static_as_background: the estimator's static report of the newest back-end frame <= f - 2
    unmasks frame f's static instances, system/main.cpp:194,217-245) and the every-2nd-frame flow of a dynamic sequence (dv_seq_input::ba_stride 2, system/main.cpp:300-312)
    on the C++ runner — tracker thread beside the estimator loop or the one-thread loop, one call or several — leave exactly what the Python pipeline leaves."""
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
    w, h, frames = 640, 360, 44
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seq = DynamicSequence(w, h, cam, frames, rate=20.0)          # default boxes: two of them are reported static for a stretch of the sequence
    kw = dict(max_cnt=150, min_dist=20, max_iters=8, use_det3d=1, static_as_background=True, ba_stride=stride)
    a, b = DynamicPipeline(seq, **kw), DynamicPipeline(seq, **kw)
    runner = Runner([a], group_size=0, threads=1)
    runner.set("tracker_thread", tracker_thread)
    assert sum(calls) == frames - 1
    for n in calls:
        runner.run(n)
    reports = 0
    for _ in range(frames - 1):
        b.step()
        reports += int(len(b.static_snaps) > 0 and len(b.static_snaps[-1][1]) > 0)
    st, poses, iters, fr = runner.get(0)
    assert np.array_equal(np.ctypeslib.as_array(st.window), b.est.window())
    want = np.array(b.poses)
    assert len(poses) == len(want) and np.array_equal(poses[:, 1:], want)
    Ia, _ = a.est.instances(); Ib, _ = b.est.instances()
    assert Ia.tobytes() == Ib.tobytes()
    assert reports >= (8 if stride == 1 else 3), reports          # the feedback really acted
    runner.close(); a.ctx.close(); b.ctx.close()


@pytest.mark.parametrize("tracker_thread,calls", [(1, (29,)), (0, (29,)), (1, (7, 1, 13, 8))])
def test_runner_dynamic_mode_equals_the_python_dynamic_pipeline(tracker_thread, calls):
    """tracker_thread 1 (default): the reference's T2 / T3 — a tracker thread filling a ring of collected frames beside the estimator loop (system/main.cpp:178-330,
    :394-404, basic/feature_queue.h); 0: the one-thread loop.  Both, and a run cut into several dv_runner_run calls (the tracker thread ends and restarts at every call,
    one frame ahead), must leave the same bits as the Python pipeline.
    dv_runner_set_dynamic: the reference's dynamic loop per frame in C++ (TrackSemanticImage + InstsTrack enqueued together, collect, the three-phase back end with
    the object branch beside the window solve) must leave EXACTLY what pipeline.DynamicPipeline leaves: ego window states, trajectory, object states, and what the
    object branch was fed — objects in every frame, extra points from the disparity map on the device"""
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
    w, h, frames = 752, 480, 30
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seq = DynamicSequence(w, h, cam, frames, rate=20.0, boxes=("escort", 4))
    kw = dict(max_cnt=150, min_dist=20, max_iters=8, use_det3d=1, mask_morphology_size=5)
    a, b = DynamicPipeline(seq, **kw), DynamicPipeline(seq, **kw)
    runner = Runner([a], group_size=0, threads=1)
    runner.set("tracker_thread", tracker_thread)
    assert sum(calls) == frames - 1
    for n in calls:
        runner.run(n)
    for _ in range(frames - 1):
        b.step()
    st, poses, iters, fr = runner.get(0)
    assert fr == frames - 1 and st.frame == b.last_state.frame and st.nonlinear == b.last_state.nonlinear
    assert np.array_equal(np.ctypeslib.as_array(st.window), b.est.window())
    want = np.array(b.poses)
    assert len(poses) == len(want) >= frames - 14 and np.array_equal(poses[:, 1:], want)
    Ia, Sa = a.est.instances(); Ib, Sb = b.est.instances()
    assert len(Ia) == len(Ib) >= 3 and Ia.tobytes() == Ib.tobytes() and np.array_equal(Sa, Sb)
    stats = runner.dynamic_stats(0)
    for key in ("object_detections", "object_features", "frames_with_objects", "min_detections"):
        assert stats[key] == b.stat[key], (key, stats, b.stat)
    assert stats["min_detections"] >= 3 and stats["object_features"] > 20 * frames
    runner.close(); a.ctx.close(); b.ctx.close()
