"""Long-run parity evidence: N frames of images -> tracker -> estimator on the HIP path (through the C ABI) against the CPU oracle on the same
rendered frames and IMU stream, frame by frame.  What the short parity tests cannot say: how often the +-1-iteration allowance of tests/conftest.py::iterations_agree is
taken over a long run, how far the states drift apart when it is, and whether the front-end hand-over stays bit-identical for a thousand frames.
usage: python tests/tools/longrun_parity.py raw|dynamic|dynamic_static <frames> [w h]      -> one JSON line
(dynamic_static: the room scene with a box nearly at rest and para::is_static_inst_as_background — the reference's default — on both sides: the estimator's static report
unmasks object pixels in the tracker two frames later, choice T1 of DESIGN.md 2) (tests/test_longrun_parity.py asserts on a shorter run of the same code)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(mode, frames, w=640, h=360, chunk=250):
    from dynamic_vins_amd import dynsim, sim
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence, Pipeline, SyntheticSequence
    from tests import oracle_py
    from tests.conftest import iterations_agree
    oracle = oracle_py.load()
    cam = sim.ZED if (w, h) == (1280, 720) else sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    max_cnt, min_dist, iters = (250, 25, 10) if (w, h) == (1280, 720) else (150, 20, 8)
    dyn = mode in ("dynamic", "dynamic_static")
    static_bg = mode == "dynamic_static"
    t_start = time.time()
    if static_bg:
        seq = DynamicSequence(w, h, cam, frames, rate=20.0)          # dynsim.default_boxes: two moving boxes, one nearly at rest
    else:
        seq = (DynamicSequence(w, h, cam, frames, rate=20.0, boxes=("escort", 3)) if dyn else SyntheticSequence(w, h, cam, frames, rate=20.0))
    pipe = (DynamicPipeline(seq, max_cnt=max_cnt, min_dist=min_dist, max_iters=iters, use_det3d=1, static_as_background=static_bg) if dyn else Pipeline(seq, max_cnt=max_cnt, min_dist=min_dist, max_iters=iters))
    snaps = []
    camt = sim.cam_tuple(cam)
    trk = oracle.tracker(w, h, max_cnt, min_dist, 1, 1, camt, camt)
    oin = oracle.insts(trk, 50, 5, 1) if dyn else None
    ekw = dict(dynamic=1, use_det3d=1, static_inst_threshold=1.0) if dyn else {}
    est = oracle.estimator(use_imu=1, stereo=1, max_iters=iters, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **ekw, **seq.noise)
    k_imu = 0
    st = dict(mode=mode, frames=frames, w=w, h=h, rows_bit_identical=0, rows_differ_first=None, solved=0, iter_equal=0, iter_plus_minus_one=0, iter_other=0, flags_differ=0,
              max_dp_m=0.0, max_dp_after_pm1_m=0.0, worst_frame=None, obj_rows=0, obj_rows_differ=0, obj_p_m=0.0, iterations_hip=0, iterations_oracle=0)
    dev_p, ref_p, pm1_frames, mism, obj_log = [], [], [], [], []
    first_above = {}
    for k in range(frames):
        t = seq.times[k]
        sd = pipe.step()
        while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
            est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
        left, right = seq.host_frame(k)
        if dyn:
            mask_o = seq.inv_mask[k]
            if static_bg:          # FeatureTrack (system/main.cpp:217-245) on the oracle side: the report of the newest back-end frame <= k - DV_STATIC_REPORT_LAG, applied on the host
                from dynamic_vins_amd import _abi, viode
                best = [sn for sn in snaps if sn[0] <= k - _abi.DV_STATIC_REPORT_LAG]
                mask_o = viode.unmask_static(seq.inv_mask[k], seq.dets[k], best[-1][1] if best else [])
                px = int(((seq.inv_mask[k] == 0) & (mask_o == 255)).sum())
                st["unmasked_px"] = st.get("unmasked_px", 0) + px; st["unmasked_frames"] = st.get("unmasked_frames", 0) + int(px > 0)
            rows_o = trk.track_image(left, right, t, mask=mask_o, mode=2, erode_k=0)
            oin.set_disparity(seq.disp_host(k), seq.baseline)
            io, fo, po = oin.track(left, right, t, seq.dets[k], seq.boxes3d[k], dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
            same_obj = len(fo) == len(pipe.ifeats) and fo.tobytes() == pipe.ifeats.tobytes() and np.array_equal(po, pipe.ipts)
            st["obj_rows"] += len(fo); st["obj_rows_differ"] += int(not same_obj)
        else:
            rows_o = trk.track_image(left, right, t)
        same = len(rows_o) == len(pipe.rows) and rows_o.tobytes() == pipe.rows.tobytes()
        st["rows_bit_identical"] += int(same)
        if not same and st["rows_differ_first"] is None:
            st["rows_differ_first"] = k
        rc, so = (est.process_dynamic(rows_o, t, io, fo, po) if dyn else est.process(rows_o, t))
        assert rc == 0
        if static_bg:
            snaps = (snaps + [(k, est.static_instances())])[-4:]
            st["static_reports_differ"] = st.get("static_reports_differ", 0) + int(not np.array_equal(snaps[-1][1], pipe.est.static_instances()))
        if (sd.frame, sd.nonlinear, sd.margin_old, sd.n_landmarks, sd.n_long) != (so.frame, so.nonlinear, so.margin_old, so.n_landmarks, so.n_long):
            st["flags_differ"] += 1
        if so.nonlinear:
            st["solved"] += 1
            st["iterations_hip"] += int(sd.iterations); st["iterations_oracle"] += int(so.iterations)
            if sd.iterations == so.iterations:
                st["iter_equal"] += 1
            elif iterations_agree(sd, so):
                st["iter_plus_minus_one"] += 1; pm1_frames.append(k)
            else:
                st["iter_other"] += 1
            if sd.iterations != so.iterations:
                mism.append(dict(frame=k, hip=int(sd.iterations), oracle=int(so.iterations), cost0_hip=float(sd.initial_cost), cost0_oracle=float(so.initial_cost), cost1_hip=float(sd.final_cost), cost1_oracle=float(so.final_cost)))
            Wd, Wo = pipe.est.window(), est.window()
            dp = float(np.abs(Wd[:, :3] - Wo[:, :3]).max())
            if dp > st["max_dp_m"]:
                st["max_dp_m"], st["worst_frame"] = dp, k
            for bar in (1e-6, 1e-5, 1e-4):
                if dp > bar and bar not in first_above:
                    first_above[bar] = k
            if pm1_frames and k - pm1_frames[-1] <= 11:          # a window still holds the frame whose solve ended an iteration apart
                st["max_dp_after_pm1_m"] = max(st["max_dp_after_pm1_m"], dp)
            dev_p.append(Wd[10, :3].copy()); ref_p.append(Wo[10, :3].copy())
            if dyn:
                Io, _ = est.instances(dynsim.INSTSTATE_DTYPE); Id, _ = pipe.est.instances()
                if len(Io) == len(Id):
                    for a, b in zip(Io, Id):
                        d_obj = float(np.abs(a["window"][:, :3] - b["window"][:, :3]).max())
                        st["obj_p_m"] = max(st["obj_p_m"], d_obj)
                        if d_obj > 1e-3 and len(obj_log) < 40:          # an object whose window leaves the oracle's: what both sides hold for it
                            ints = ["id", "is_initial", "is_tracking", "is_curr_visible", "is_static", "is_init_velocity", "age", "lost_number", "n_landmarks", "n_valid", "triangle_num"]
                            obj_log.append(dict(frame=k, dp=d_obj, oracle={f: int(a[f]) for f in ints}, hip={f: int(b[f]) for f in ints},
                                                p_oracle=a["window"][-1, :3].tolist(), p_hip=b["window"][-1, :3].tolist(),
                                                obj_summary_hip=[int(pipe.est.obj_last.iterations), float(pipe.est.obj_last.initial_cost), float(pipe.est.obj_last.final_cost)] if hasattr(pipe.est, "obj_last") else None))
                else:
                    st["flags_differ"] += 1
    dev_p, ref_p = np.array(dev_p), np.array(ref_p)
    st["ate_hip_vs_oracle_m"] = float(sim.align_ate(dev_p, ref_p)[0])
    st["max_abs_traj_diff_m"] = float(np.abs(dev_p - ref_p).max())
    st["ate_hip_vs_ground_truth_m"] = float(pipe.ate())
    st["pm1_frames"] = pm1_frames[:50]
    st["iteration_mismatches"] = mism[:50]
    st["objects_above_1mm"] = obj_log
    st["first_frame_with_window_deviation_above"] = {"%g" % b: f for b, f in sorted(first_above.items())}
    st["wall_s"] = round(time.time() - t_start, 1)
    pipe.ctx.close()
    return st


if __name__ == "__main__":
    mode, frames = sys.argv[1], int(sys.argv[2])
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (640, 360)
    print(json.dumps(run(mode, frames, w, h)))
