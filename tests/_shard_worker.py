"""worker of tests/test_sharded_solve.py (GPU box):  python -m torch.distributed.run --nproc-per-node 2 tests/_shard_worker.py OUT
Two processes share the one GPU; the exchange vectors travel through the host transport (pinned memory + gloo all-gather) or, argv[2] = peer, through the
one-shot peer transport (each rank writes into the window the other exposes through hipIpc).  Each rank solves every
problem twice — sharded (its ctx is one rank of the window) and unsharded (a second, plain ctx) — and writes both for the parent to compare."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from dynamic_vins_amd import dist as dv_dist, sim
from dynamic_vins_amd.backend import Estimator, ba_eval, ba_solve
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py

rank, world, _ = dv_dist.init(prefer_gpu=False)        # gloo: the two ranks share one device, RCCL refuses that
oracle = oracle_py.load()                               # ba_gen pre-integrates the IMU factors with the oracle's integrator (test infrastructure)
shard = Context(width=64, height=64, max_cnt=10, min_dist=5)
plain = Context(width=64, height=64, max_cnt=10, min_dist=5)
dv_dist.shard_window(shard, rank, world, transport=sys.argv[2] if len(sys.argv) > 2 else "host")
out = {}

CASES = [dict(seed=2, with_prior=True), dict(seed=6, nlm=300, max_iters=10, with_prior=True), dict(seed=3, use_imu=0, nframes=7), dict(seed=9, nlm=1, max_iters=3),
         dict(seed=12, nlm=0, max_iters=4, with_prior=True), dict(seed=4, with_prior=True, outlier_ratio=0.1, max_iters=10), dict(seed=11, nlm=1000, max_iters=4, with_prior=True)]
for i, kw in enumerate(CASES):
    ref = ba_gen.make_window(oracle, **kw)
    a, b = ref.clone(), ref.clone()
    ca, Sa, ga = ba_eval(shard, a)
    cb, Sb, gb = ba_eval(plain, b)
    out[f"eval{i}_S"] = np.array([np.abs(Sa - Sb).max() / max(np.abs(Sb).max(), 1e-300), np.abs(ga - gb).max() / max(np.abs(gb).max(), 1e-300), abs(ca - cb) / max(abs(cb), 1e-300)])
    out[f"eval{i}_digest"] = np.array([Sa.sum(), ga.sum(), ca])
    sa, sb = ba_solve(shard, a), ba_solve(plain, b)
    out[f"solve{i}_sum"] = np.array([sa.iterations, sb.iterations, sa.termination, sb.termination, sa.final_cost, sb.final_cost, sa.initial_cost, sb.initial_cost])
    out[f"solve{i}_shard"] = np.concatenate([a.pose.ravel(), a.speed_bias.ravel(), a.inv_depth.ravel()])
    out[f"solve{i}_plain"] = np.concatenate([b.pose.ravel(), b.speed_bias.ravel(), b.inv_depth.ravel()])

# the estimator's fused path (solve + gauge fix + marginalization enqueued back to back, prior resident in HBM) on a sharded window
NOISE = dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
traj = sim.Trajectory()
fs = sim.FeatureSim(traj, sim.EUROC, 752, 480, sim.room_points(3000), max_cnt=150, pix_sigma=0.3, seed=3)
kw = dict(use_imu=1, stereo=1, max_iters=8, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **NOISE)
ea, eb = Estimator(shard, **kw), Estimator(plain, **kw)
frames, T0, dtf = 30, 1.0, 0.1
ts, acc, gyr = sim.imu_stream(traj, T0 - 0.05, T0 + frames * dtf + 0.1, 200.0, **NOISE)
k = 0
Wa, Wb, its = [], [], []
for f in range(frames):
    t = T0 + f * dtf
    while k < len(ts) and ts[k] <= t + 0.011:
        ea.InputIMU(ts[k], acc[k], gyr[k]); eb.InputIMU(ts[k], acc[k], gyr[k])
        k += 1
    rows = fs.frame(t)
    _, sa = ea.ProcessMeasurements(rows, t)
    _, sb = eb.ProcessMeasurements(rows, t)
    Wa.append(ea.window().copy()); Wb.append(eb.window().copy()); its.append([sa.iterations, sb.iterations, sa.n_landmarks, sb.n_landmarks])
out["est_shard"], out["est_plain"], out["est_its"] = np.array(Wa), np.array(Wb), np.array(its)

# the operator form of the exchange
v = np.arange(1000, dtype=np.float64) * (1.0 + rank) + 0.1 * rank
assert shard.lib.dv_allreduce_reduced_system(shard.h, v.ctypes.data, len(v)) == 0
out["allreduce"] = v
info = dv_dist.dist_info(shard)
out["info"] = np.array([info["rank"], info["world"], info["exchanges"]])
np.savez(os.path.join(sys.argv[1], f"shard_rank{rank}.npz"), **out)
dv_dist.barrier()
shard.close(); plain.close()
dv_dist.finalize()
