"""Fixtures for tests/test_obj_sequence_problems.py: the object-solve problems (InstanceManager::Optimization, estimator_insts.cpp:772-807) that a REAL dynamic sequence
produces, as they enter the solve — CPU only, from the oracle's estimator (hook "obj_dump" of oracle/inst_manager.h).  The random problems of tests/obj_gen.py are well
posed; a sequence is not always: in the escort scene one box is classified static for ~25 frames while it travels with the camera, its enclose factors sit far outside the
hinge and the solve stagnates (initial cost == final cost) — the regime in which the long dynamic run (tests/tools/longrun_parity.py dynamic 600 1280 720) sees the object
states of HIP and oracle part.  usage: python tests/tools/obj_problem_dump.py -> tests/golden/obj_sequence_problems.npz"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
KEEP = (60, 90, 103, 106, 109, 114, 120, 124, 127, 131)


def main():
    from dynamic_vins_amd import dynsim, sim
    from dynamic_vins_amd.backend import OBJBOX_DTYPE, OBJPT_DTYPE
    from dynamic_vins_amd.pipeline import DynamicSequence
    from tests import oracle_py
    oracle = oracle_py.load()
    w, h, frames = 1280, 720, max(KEEP) + 1
    seq = DynamicSequence(w, h, sim.ZED, frames, rate=20.0, boxes=("escort", 3))
    camt = sim.cam_tuple(sim.ZED)
    trk = oracle.tracker(w, h, 250, 25, 1, 1, camt, camt)
    oin = oracle.insts(trk, 50, 5, 1)
    est = oracle.estimator(use_imu=1, stereo=1, max_iters=10, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], dynamic=1, use_det3d=1, static_inst_threshold=1.0, **seq.noise)
    tmp = tempfile.mkdtemp(prefix="dvo_obj_")
    os.environ["DVO_OBJ_DUMP_DIR"] = tmp
    k_imu, out = 0, {}
    for k in range(frames):
        t = seq.times[k]
        while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
            est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
        left, right = seq.host_frame(k)
        rows = trk.track_image(left, right, t, mask=seq.inv_mask[k], mode=2, erode_k=0)
        oin.set_disparity(seq.disp_host(k), seq.baseline)
        io, fo, po = oin.track(left, right, t, seq.dets[k], seq.boxes3d[k], dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
        before = set(os.listdir(tmp))
        oracle.lib.dvo_set_variant(b"obj_dump", 1 if k in KEEP else 0)
        rc, so = est.process_dynamic(rows, t, io, fo, po)
        oracle.lib.dvo_set_variant(b"obj_dump", 0)
        assert rc == 0
        new = sorted(set(os.listdir(tmp)) - before)
        if not new:
            continue
        raw = open(os.path.join(tmp, new[-1]), "rb").read()
        hdr = np.frombuffer(raw, np.int32, 6)
        n_obj, n_boxes, n_points, max_iters, plane_kind = (int(x) for x in hdr[:5])
        off = 24
        def take(dtype, n):
            nonlocal off
            a = np.frombuffer(raw, dtype, n, off).copy(); off += a.nbytes
            return a
        state, dims, body, R_bc = take(np.float64, n_obj * 77), take(np.float64, n_obj * 3), take(np.float64, 77), take(np.float64, 9)
        boxes, points = take(OBJBOX_DTYPE, n_boxes), take(OBJPT_DTYPE, n_points)
        assert off == len(raw)
        Io, summ = est.instances(dynsim.INSTSTATE_DTYPE)
        out.update({f"f{k}_state": state.reshape(n_obj, 11, 7), f"f{k}_dims": dims.reshape(n_obj, 3), f"f{k}_body": body.reshape(11, 7), f"f{k}_R_bc": R_bc, f"f{k}_boxes": boxes,
                    f"f{k}_points": points, f"f{k}_meta": np.array([max_iters, plane_kind]), f"f{k}_static": np.array([int(a["is_static"]) for a in Io]),
                    f"f{k}_summary": np.asarray(summ, np.float64)})
        print(k, "objects", n_obj, "boxes", n_boxes, "points", n_points, "static", out[f"f{k}_static"].tolist(), "summary", [round(float(x), 4) for x in summ], flush=True)
    out["frames"] = np.array(sorted(int(k[1:].split("_")[0]) for k in out if k.endswith("_state")))
    path = os.path.join(ROOT, "tests", "golden", "obj_sequence_problems.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
