import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import oracle_py
from dynamic_vins_amd import sim
from dynamic_vins_amd.frontend import Context
from dynamic_vins_amd.backend import Estimator
o = oracle_py.load()
use_imu = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
traj = sim.Trajectory(); pts = sim.room_points(3000)
fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, pts, max_cnt=150, pix_sigma=0.3, seed=3)
kw = dict(use_imu=use_imu, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
ref = o.estimator(**kw); dev = Estimator(ctx, **kw)
T0, dtf = 1.0, 0.1
ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE)
k = 0
for f in range(frames):
    t = T0 + f * dtf
    while k < len(ts) and ts[k] <= t + 0.011:
        ref.input_imu(ts[k], acc[k], gyr[k]); dev.InputIMU(ts[k], acc[k], gyr[k]); k += 1
    rows = fs.frame(t)
    rc_o, so = ref.process(rows, t); rc_d, sd = dev.ProcessMeasurements(rows, t)
    Wo, Wd = ref.window(), dev.window()
    print(f, "nl", so.nonlinear, sd.nonlinear, "mo", so.margin_old, sd.margin_old, "it", so.iterations, sd.iterations,
          "cost0 %.6f %.6f" % (so.initial_cost, sd.initial_cost), "cost %.6f %.6f" % (so.final_cost, sd.final_cost),
          "dP %.2e dQ %.2e dV %.2e dB %.2e" % (np.abs(Wo[:, :3]-Wd[:, :3]).max(), np.abs(Wo[:, 3:7]-Wd[:, 3:7]).max(), np.abs(Wo[:, 7:10]-Wd[:, 7:10]).max(), np.abs(Wo[:, 10:]-Wd[:, 10:]).max()))
