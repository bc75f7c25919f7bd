"""debug: where do the object states of the HIP estimator leave the oracle's in a long dynamic 1280x720 run?  (tests/tools/longrun_parity.py dynamic 600 1280 720 saw 4.6 m)
usage: python tests/tools/obj_diverge.py <frames> <first_logged_frame>"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamic_vins_amd import dynsim, sim
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
from tests import oracle_py

frames, first = int(sys.argv[1]), int(sys.argv[2])
w, h = 1280, 720
oracle = oracle_py.load()
cam = sim.ZED
seq = DynamicSequence(w, h, cam, frames, rate=20.0, boxes=("escort", 3))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, use_det3d=1)
camt = sim.cam_tuple(cam)
trk = oracle.tracker(w, h, 250, 25, 1, 1, camt, camt)
oin = oracle.insts(trk, 50, 5, 1)
est = oracle.estimator(use_imu=1, stereo=1, max_iters=10, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], dynamic=1, use_det3d=1, static_inst_threshold=1.0, **seq.noise)
k_imu = 0
for k in range(frames):
    t = seq.times[k]
    sd = pipe.step()
    while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
        est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
    left, right = seq.host_frame(k)
    rows_o = trk.track_image(left, right, t, mask=seq.inv_mask[k], mode=2, erode_k=0)
    oin.set_disparity(seq.disp_host(k), seq.baseline)
    io, fo, po = oin.track(left, right, t, seq.dets[k], seq.boxes3d[k], dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
    rc, so = est.process_dynamic(rows_o, t, io, fo, po)
    if k < first:
        continue
    Io, so4 = est.instances(dynsim.INSTSTATE_DTYPE); Id, sd4 = pipe.est.instances()
    Wd, Wo = pipe.est.window(), est.window()
    line = dict(frame=k, ego_dp=float(np.abs(Wd[:, :3] - Wo[:, :3]).max()), sum_oracle=[float(x) for x in so4], sum_hip=[float(x) for x in sd4], objs=[])
    for a, b in zip(Io, Id):
        dwin = np.abs(a["window"][:, :3] - b["window"][:, :3]).max(axis=1)
        o = dict(id=int(a["id"]), static=(int(a["is_static"]), int(b["is_static"])), static_frame=(int(a["static_frame"]), int(b["static_frame"])), tri=(int(a["triangle_num"]), int(b["triangle_num"])),
                 nlm=(int(a["n_landmarks"]), int(b["n_landmarks"])), dwin=["%.1e" % x for x in dwin], dvel=float(np.abs(a["vel_v"] - b["vel_v"]).max()), dvela=float(np.abs(a["vel_a"] - b["vel_a"]).max()),
                 ddims=float(np.abs(a["dims"] - b["dims"]).max()), vel_o=[float(x) for x in a["vel_v"]])
        line["objs"].append(o)
    print(json.dumps(line))
