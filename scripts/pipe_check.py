"""end-to-end check: rendered stereo + IMU -> HIP front end -> HIP back end; ATE vs ground truth; timing"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_vins_amd import sim, render
from dynamic_vins_amd.frontend import Context, make_cam, DV_MEM_DEVICE
from dynamic_vins_amd.backend import Estimator
W, H = (1280, 720) if len(sys.argv) < 2 or sys.argv[1] == "720" else (752, 480)
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cam = sim.ZED if W == 1280 else sim.EUROC
max_cnt, min_dist = (250, 25) if W == 1280 else (150, 30)
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
traj = sim.Trajectory()
t0 = time.time()
rr = render.RoomRenderer(cam, W, H)
T0, dtf = 1.0, 0.05
frames = [rr.stereo(traj, T0 + k * dtf) for k in range(NF)]
torch.cuda.synchronize()
print("render %.2f s for %d stereo frames" % (time.time() - t0, NF), "mean gray", float(frames[0][0].float().mean()))
ctx = Context(width=W, height=H, max_cnt=max_cnt, min_dist=min_dist, cam0=make_cam(*sim.cam_tuple(cam)), cam1=make_cam(*sim.cam_tuple(cam)))
est = Estimator(ctx, use_imu=1, stereo=1, max_iters=10, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + NF * dtf + 0.1, 200.0, **NOISE)
k = 0
est_p, gt_p = [], []
ctx.timing_enable(True)
tt = time.perf_counter(); t_track = 0; t_ba = 0
for f in range(NF):
    t = T0 + f * dtf
    while k < len(ts) and ts[k] <= t + 0.006:
        est.InputIMU(ts[k], acc[k], gyr[k]); k += 1
    a = time.perf_counter()
    rows = ctx.track_stereo(frames[f][0].data_ptr(), frames[f][1].data_ptr(), t, mem=DV_MEM_DEVICE)
    b = time.perf_counter()
    rc, st = est.ProcessMeasurements(rows, t)
    c = time.perf_counter()
    t_track += b - a; t_ba += c - b
    assert rc == 0
    if st.nonlinear:
        est_p.append(est.window()[10, :3].copy()); gt_p.append(traj.p(t))
    if f % 10 == 0 or f < 3:
        print(f, "feats", len(rows), "tracked", int((rows["track_cnt"] > 1).sum()), "stereo", int(rows["has_right"].sum()), "lms", st.n_landmarks, st.n_long, "it", st.iterations, "cost %.1f->%.1f" % (st.initial_cost, st.final_cost), "mo", st.margin_old)
wall = time.perf_counter() - tt
print("wall %.3f s, %.2f ms/frame (%.0f fps) track %.2f ms ba %.2f ms" % (wall, wall / NF * 1e3, NF / wall, t_track / NF * 1e3, t_ba / NF * 1e3))
rmse, _, _ = sim.align_ate(est_p, gt_p)
print("ATE rmse vs ground truth: %.4f m over %d poses" % (rmse, len(est_p)))
for name in ["h_imu", "h_process_total", "h_add_features", "h_triangulate", "h_build", "h_solve_total", "h_solve_upload", "h_solve_enqueue", "h_post", "h_marg_total", "h_reject", "h_slide", "frame", "pyr", "lk_temporal", "compact", "gftt_eig", "gftt_select", "lk_stereo", "finalize", "ba_solve", "ba_marg"]:
    ms, cnt = ctx.timing_get(name)
    if cnt: print(f"  {name:12s} {ms/cnt*1e3:9.1f} us  (n={cnt})")
