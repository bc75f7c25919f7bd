"""frame-by-frame iteration counts / costs / ego deviation of the dynamic 640x360 test sequence: product (MF16 or a debug form) against the oracle.
usage: dyn_iter_trace.py [debug key]   (tests/ territory: uses the oracle)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamic_vins_amd import dynsim, sim                                     # noqa: E402
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence       # noqa: E402
from tests import oracle_py                                                   # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else ""
o = oracle_py.load()
w, h, frames = 640, 360, 40
cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
seq = DynamicSequence(w, h, cam, frames, rate=20.0)
pipe = DynamicPipeline(seq, max_cnt=150, min_dist=20, max_iters=8)
if key:
    assert pipe.ctx.lib.dv_debug_set(pipe.ctx.h, key.encode(), 1) == 0
camt = sim.cam_tuple(cam)
trk = o.tracker(w, h, 150, 20, 1, 1, camt, camt)
oin = o.insts(trk, 50, 5, 1)
est = o.estimator(use_imu=1, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], dynamic=1, use_det3d=1, static_inst_threshold=1.0, **seq.noise)
k_imu = 0
for k in range(frames):
    t = seq.times[k]
    sd = pipe.step()
    while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
        est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
    left, right = seq.host_frame(k)
    rows_o = trk.track_image(left, right, t, mask=seq.inv_mask[k], mode=2)
    io, fo, po = oin.track(left, right, t, seq.dets[k], seq.boxes3d[k], dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
    rc, so = est.process_dynamic(rows_o, t, io, fo, po)
    dp = np.abs(pipe.est.window()[:, :3] - est.window()[:, :3]).max()
    print("frame %2d it %d/%d cost0 %.9g/%.9g cost %.9g/%.9g  dp %.3g" % (k, sd.iterations, so.iterations, sd.initial_cost, so.initial_cost, sd.final_cost, so.final_cost, dp))
