"""debug: two ranks on one GPU, host transport: exchange counts and solve summaries per case (python -m torch.distributed.run --nproc-per-node 2 scripts/dbg/shard_dbg_worker.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from dynamic_vins_amd import dist as dv_dist
from dynamic_vins_amd.backend import ba_eval, ba_solve
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py
rank, world, _ = dv_dist.init(prefer_gpu=False)
oracle = oracle_py.load()
shard = Context(width=64, height=64, max_cnt=10, min_dist=5)
plain = Context(width=64, height=64, max_cnt=10, min_dist=5)
dv_dist.shard_window(shard, rank, world, transport="host")
CASES = [dict(seed=2, with_prior=True), dict(seed=6, nlm=300, max_iters=10, with_prior=True), dict(seed=3, use_imu=0, nframes=7), dict(seed=9, nlm=1, max_iters=3),
         dict(seed=12, nlm=0, max_iters=4, with_prior=True), dict(seed=4, with_prior=True, outlier_ratio=0.1, max_iters=10), dict(seed=11, nlm=1000, max_iters=4, with_prior=True)]
for i, kw in enumerate(CASES):
    ref = ba_gen.make_window(oracle, **kw)
    a, b = ref.clone(), ref.clone()
    e0 = dv_dist.dist_info(shard)["exchanges"]
    ca, Sa, ga = ba_eval(shard, a)
    cb, Sb, gb = ba_eval(plain, b)
    e1 = dv_dist.dist_info(shard)["exchanges"]
    print(f"[rank {rank}] case {i} eval: exchanges {e1 - e0}, dS {np.abs(Sa - Sb).max() / max(np.abs(Sb).max(), 1e-300):.2e} dg {np.abs(ga - gb).max() / max(np.abs(gb).max(), 1e-300):.2e} dc {abs(ca - cb) / max(abs(cb), 1e-300):.2e}", flush=True)
    sa = ba_solve(shard, a)
    e2 = dv_dist.dist_info(shard)["exchanges"]
    sb = ba_solve(plain, b)
    print(f"[rank {rank}] case {i} solve: exchanges {e2 - e1}, iterations {sa.iterations}/{sb.iterations} term {sa.termination}/{sb.termination} cost {sa.final_cost:.12g}/{sb.final_cost:.12g} "
          f"dpose {np.abs(a.pose - b.pose).max():.2e} ddepth {np.abs(a.inv_depth - b.inv_depth).max(initial=0):.2e}", flush=True)
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Estimator
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
traj = sim.Trajectory()
fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(3000), max_cnt=150, pix_sigma=0.3, seed=3)
kw = dict(use_imu=1, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
ea, eb = Estimator(shard, **kw), Estimator(plain, **kw)
frames, T0, dtf = 30, 1.0, 0.1
ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE)
k = 0
for f in range(frames):
    t = T0 + f * dtf
    while k < len(ts) and ts[k] <= t + 0.011:
        ea.InputIMU(ts[k], acc[k], gyr[k]); eb.InputIMU(ts[k], acc[k], gyr[k])
        k += 1
    rows = fs.frame(t)
    e0 = dv_dist.dist_info(shard)["exchanges"]
    _, sa = ea.ProcessMeasurements(rows, t)
    e1 = dv_dist.dist_info(shard)["exchanges"]
    _, sb = eb.ProcessMeasurements(rows, t)
    print(f"[rank {rank}] frame {f}: exchanges {e1 - e0} iterations {sa.iterations}/{sb.iterations} landmarks {sa.n_landmarks}/{sb.n_landmarks} dwin {np.abs(ea.window()[:, :7] - eb.window()[:, :7]).max():.2e}", flush=True)
dv_dist.barrier()
shard.close(); plain.close()
dv_dist.finalize()
