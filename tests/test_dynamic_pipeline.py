"""Dynamic mode end to end (`-m gpu`): rendered stereo frames with 3 moving boxes (dynamic variant) through
  HIP:    dv_track_stereo (TrackSemanticImage) + dv_inst_track (InstsFeatManager) -> dv_est_process_dynamic (ProcessImage with the object branch)
  oracle: dvo_tracker (mode 2) + dvo_insts -> dvo_estimator_process_dynamic
on the same frames, masks, detections and IMU stream.  Bars: background rows AND object rows bit-identical every frame (ids, track counts, fp64 bit
patterns of the normalised points / velocities), object tables identical (ids, 3-D box association, extra points), estimator flags / counts identical,
ego window within 1e-5 m, object states within the tolerances of tests/test_dynamic_parity.py."""
import numpy as np

from dynamic_vins_amd import _abi
import pytest

from tests.conftest import iterations_agree
from dynamic_vins_amd import dynsim, sim

pytestmark = pytest.mark.gpu

INT_FIELDS = ["id", "is_initial", "is_tracking", "is_curr_visible", "is_static", "is_init_velocity", "age", "lost_number", "static_frame", "n_landmarks", "n_valid", "triangle_num"]


def rows_equal(a, b, what):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)} rows"
    assert np.array_equal(a["id"], b["id"]) and np.array_equal(a["track_cnt"], b["track_cnt"]) and np.array_equal(a["has_right"], b["has_right"]), what
    assert np.array_equal(a["left"].view(np.uint64), b["left"].view(np.uint64)), what
    assert np.array_equal(a["right"].view(np.uint64), b["right"].view(np.uint64)), what


def run(oracle, w, h, frames, max_cnt, min_dist, iters, use_det3d=1, morph=0, drop=(), static_bg=False):
    from dynamic_vins_amd import viode
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
    cam = sim.ZED if (w, h) == (1280, 720) else sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seq = DynamicSequence(w, h, cam, frames, rate=20.0)
    for k in drop:                       # frames where the detector saw nothing
        seq.dets[k], seq.boxes3d[k] = [], np.zeros(0, dynsim.BOX3D_DTYPE)
    pipe = DynamicPipeline(seq, max_cnt=max_cnt, min_dist=min_dist, max_iters=iters, use_det3d=use_det3d, mask_morphology_size=morph, static_as_background=static_bg)
    snaps = []
    camt = sim.cam_tuple(cam)
    trk = oracle.tracker(w, h, max_cnt, min_dist, 1, 1, camt, camt)
    oin = oracle.insts(trk, 50, 5, use_det3d)
    est = oracle.estimator(use_imu=1, stereo=1, max_iters=iters, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], dynamic=1, use_det3d=use_det3d, static_inst_threshold=1.0, **seq.noise)
    k_imu = 0
    stats = dict(obj_rows=0, objs=set(), initial=0, solved=0, max_dp=0.0, obj_p=0.0, obj_q=0.0, iter_mismatch=0, unmasked_px=0, unmasked_frames=0, bg_rows_on_objects=0)
    for k in range(frames):
        t = seq.times[k]
        sd = pipe.step()
        while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
            est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
        left, right = seq.host_frame(k)
        mask_o = seq.inv_mask[k]
        if static_bg:          # FeatureTrack (system/main.cpp:217-245) on the oracle side: the estimator's report of the newest back-end frame <= k - 2, applied on the host
            best = [sn for sn in snaps if sn[0] <= k - _abi.DV_STATIC_REPORT_LAG]
            mask_o = viode.unmask_static(seq.inv_mask[k], seq.dets[k], best[-1][1] if best else [])
            px = int(((seq.inv_mask[k] == 0) & (mask_o == 255)).sum())
            stats["unmasked_px"] += px; stats["unmasked_frames"] += int(px > 0)
        rows_o = trk.track_image(left, right, t, mask=mask_o, mode=2, erode_k=morph)
        if static_bg and len(rows_o):          # background features that sit on (unmasked) object pixels
            u, v = np.rint(rows_o["left"][:, 3]).astype(int).clip(0, w - 1), np.rint(rows_o["left"][:, 4]).astype(int).clip(0, h - 1)
            stats["bg_rows_on_objects"] += int((seq.inv_mask[k][v, u] == 0).sum())
        oin.set_disparity(seq.disp_host(k), seq.baseline)          # the extra points: DetectExtraPoints + ProcessExtraPoints from the same disparity map on both sides
        io, fo, po = oin.track(left, right, t, seq.dets[k], seq.boxes3d[k] if use_det3d else None, dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
        rows_equal(pipe.rows, rows_o, f"frame {k} background")
        assert len(io) == len(pipe.insts), f"frame {k}: {len(io)} vs {len(pipe.insts)} objects"
        for name in ["id", "has_box3d", "first_feat", "n_feats", "first_point", "n_points"]:
            assert np.array_equal(io[name], pipe.insts[name]), f"frame {k}: {name} {io[name]} vs {pipe.insts[name]}"
        assert np.array_equal(io["rect"], pipe.insts["rect"]) and io["box3d"].tobytes() == pipe.insts["box3d"].tobytes(), f"frame {k}"
        rows_equal(pipe.ifeats, fo, f"frame {k} objects")
        assert np.array_equal(po, pipe.ipts)
        stats["obj_rows"] += len(fo); stats["objs"].update(int(i) for i in io["id"])
        rc, so = est.process_dynamic(rows_o, t, io, fo, po)
        assert rc == 0
        if static_bg:
            snaps = (snaps + [(k, est.static_instances())])[-4:]
            assert np.array_equal(snaps[-1][1], pipe.est.static_instances()), f"frame {k}: static report {snaps[-1][1]} vs {pipe.est.static_instances()}"
        assert (sd.frame, sd.nonlinear, sd.margin_old, sd.n_landmarks, sd.n_long) == (so.frame, so.nonlinear, so.margin_old, so.n_landmarks, so.n_long), f"frame {k}"
        assert iterations_agree(sd, so), f"frame {k}: iterations {sd.iterations} vs {so.iterations}, costs {sd.initial_cost} vs {so.initial_cost}"
        stats["iter_mismatch"] += int(sd.iterations != so.iterations)
        stats["max_dp"] = max(stats["max_dp"], np.abs(pipe.est.window()[:, :3] - est.window()[:, :3]).max())
        Io, So = est.instances(dynsim.INSTSTATE_DTYPE)
        Id, Sd = pipe.est.instances()
        assert len(Io) == len(Id)
        for name in INT_FIELDS:
            assert np.array_equal(Io[name], Id[name]), f"frame {k}: {name} {Io[name]} vs {Id[name]}"
        assert So[0] == Sd[0] and So[1] == Sd[1], f"frame {k}: {So} vs {Sd}"
        stats["solved"] += int(So[0] > 0)
        for a, b in zip(Io, Id):
            stats["initial"] += int(a["is_initial"])
            stats["obj_p"] = max(stats["obj_p"], np.abs(a["window"][:, :3] - b["window"][:, :3]).max())
            qa, qb = a["window"][:, 3:], b["window"][:, 3:]
            sg = np.sign((qa * qb).sum(1, keepdims=True)); sg[sg == 0] = 1
            stats["obj_q"] = max(stats["obj_q"], np.abs(qa - sg * qb).max())
    ate = pipe.ate()
    pipe.ctx.close()
    return stats, ate


def bars(stats):
    """(ego, object position, object quaternion) bars of a run.  Strict while every window solve took the oracle's number of iterations.  When a solve ended ONE
    iteration apart (conftest.iterations_agree: the prior's constant c0 = b'^T A'^+ b' carries O(1) rounding noise — DESIGN.md M2 — and moves only Ceres' RELATIVE
    function-tolerance test; which frame it hits depends on the last bits of the factorisation, so it moved when be_solve went to the 16-wide MFMA form), the
    states differ by that last step — ~1e-5 m near convergence — and the objects, placed relative to the ego poses, follow.  North_star's bar is 1e-3 m ATE."""
    return (1e-5, 1e-5, 1e-6) if stats["iter_mismatch"] == 0 else (3e-5, 1e-4, 2e-5)


def test_dynamic_pipeline_matches_oracle_640(oracle):
    stats, ate = run(oracle, 640, 360, 40, 150, 20, 8)
    assert stats["obj_rows"] > 500 and len(stats["objs"]) >= 2 and stats["initial"] > 10 and stats["solved"] > 5, stats
    b = bars(stats)
    assert stats["max_dp"] < b[0] and stats["obj_p"] < b[1] and stats["obj_q"] < b[2], stats
    assert stats["iter_mismatch"] <= 2, stats
    assert ate < 0.05


def test_dynamic_pipeline_bench_workload_1280(oracle):
    """the bench's dynamic workload: 1280x720, max_cnt 250 / min_dist 25, 10 iterations; with detector drop-outs and in-tracker mask erosion"""
    stats, ate = run(oracle, 1280, 720, 30, 250, 25, 10, morph=5, drop=(17, 18, 24))
    assert stats["obj_rows"] > 500 and stats["solved"] > 3, stats
    b = bars(stats)
    assert stats["max_dp"] < b[0] and stats["obj_p"] < max(b[1], 5e-5) and stats["obj_q"] < max(b[2], 5e-6), stats      # (round 2: one window solve of this run ended an iteration apart: ego 6e-6 m, objects follow)


def test_static_instances_leave_the_merged_mask(oracle):
    """para::is_static_inst_as_background (vio_parameters.h:86: the reference's DEFAULT): FeatureTrack takes the pixels of the instances the estimator reported static out of
    the merged mask before TrackSemanticImage (system/main.cpp:194,217-245; InstanceManager::SetOutputInstInfo, estimator_insts.cpp:967-990 behind PushBack).  The scene's
    slow boxes are reported static for a while; their pixels are unmasked with the runner's deterministic two-frame lag, the background tracker then picks features on them.
    dv_est_get_static_instances + dv_track_unmask_static (a kernel on the tracking stream) against the oracle's report + a host-side unmask: reports identical every frame,
    background and object rows bit-identical, the usual state bars."""
    stats, ate = run(oracle, 640, 360, 44, 150, 20, 8, static_bg=True)
    assert stats["unmasked_frames"] >= 8 and stats["unmasked_px"] > 50000 and stats["bg_rows_on_objects"] > 20, stats
    b = bars(stats)
    assert stats["max_dp"] < b[0] and stats["obj_p"] < b[1] and stats["obj_q"] < b[2], stats
    assert stats["iter_mismatch"] <= 2, stats


def test_dynamic_pipeline_without_det3d(oracle):
    stats, ate = run(oracle, 640, 360, 30, 150, 20, 8, use_det3d=0)
    b = bars(stats)
    assert stats["max_dp"] < b[0] and stats["obj_p"] < b[1], stats


@pytest.mark.parametrize("blank_key", [False, True])
def test_viode_right_key_test_rows_bit_identical(oracle, blank_key):
    """cfg::dataset == kViode: InstFeat::TrackRightByPad keeps a right-image point only where seg1 carries the object's key (front_end/instance_feature.cpp:263-268).
    dv_inst_set_right_keys (a kernel behind the stereo LK of the objects) against the oracle's dvo_insts_set_right_keys: background and object rows bit-identical every
    frame; with one object's key wiped from seg1 that object keeps NO right observation on either side while the others keep theirs."""
    from dynamic_vins_amd import viode
    from dynamic_vins_amd.frontend import Context, DV_MEM_DEVICE, DV_MODE_SEMANTIC, make_cam
    w, h, frames = 640, 360, 10
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    c = make_cam(*sim.cam_tuple(cam))
    ctx = Context(width=w, height=h, max_cnt=150, min_dist=20, cam0=c, cam1=c, mask_morphology_size=5)
    seq = viode.ViodeSequence(w, h, cam, frames, ctx, rate=20.0)
    ctx.inst_config(50, 5, 0)
    victim = int(seq.dyn_keys[1])
    camt = sim.cam_tuple(cam)
    trk = oracle.tracker(w, h, 150, 20, 1, 1, camt, camt)
    oin = oracle.insts(trk, 50, 5, 0)
    right_obs = {int(k): 0 for k in seq.dyn_keys}
    for k in range(frames):
        t = seq.times[k]
        keys = seq.right_keys[k].copy()
        if blank_key:
            keys[keys == victim] = 12345
        l, r = seq.frames[k]
        ctx.track_stereo_enqueue(l.data_ptr(), r.data_ptr(), t, seq.inv_mask_dev[k].data_ptr(), DV_MODE_SEMANTIC, DV_MEM_DEVICE)
        ctx.inst_set_right_keys(keys)
        ctx.inst_track_enqueue(t, seq.dets[k], None)
        rows = ctx.track_stereo_collect()
        insts, ifeats, pts = ctx.inst_track_collect()
        left, right = seq.host_frame(k)
        rows_o = trk.track_image(left, right, t, mask=seq.inv_mask[k], mode=2, erode_k=5)
        oin.set_right_keys(keys)
        io, fo, po = oin.track(left, right, t, seq.dets[k], None, dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
        rows_equal(rows, rows_o, f"frame {k} background")
        assert np.array_equal(io["id"], insts["id"]) and np.array_equal(io["n_feats"], insts["n_feats"]), k
        rows_equal(ifeats, fo, f"frame {k} objects")
        for o in io:
            f = fo[o["first_feat"]: o["first_feat"] + o["n_feats"]]
            right_obs[int(o["id"])] += int(f["has_right"].sum())
    ctx.close()
    others = [v for k_, v in right_obs.items() if k_ != victim]
    assert min(others) > 5 * frames, right_obs
    assert (right_obs[victim] == 0) if blank_key else (right_obs[victim] > 5 * frames), right_obs
