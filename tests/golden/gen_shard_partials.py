"""Generates tests/golden/shard_partials.npz ON THE GPU BOX (python tests/golden/gen_shard_partials.py OUT.npz): the per-rank partial reduced camera
systems [S | g | cost] of one window sharded by landmark over 2 and over 4 ranks, produced by the HIP path (dv_ba_eval on each rank's share: its
landmarks; IMU factors and prior on rank 0 only, as include/dvins.h describes the exchange) and the unsharded system of the same window.
tests/test_distributed.py replays them through the gloo all-gather + rank-ordered sum on CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from dynamic_vins_amd import dist as dv_dist
from dynamic_vins_amd.backend import WindowProblem, ba_eval
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py


def share(prob, rank, world):
    mine = dv_dist.shard_landmarks(len(prob.landmarks), rank, world)
    facs, lms = [], []
    for k, l in enumerate(mine):
        L = prob.landmarks[l]
        first = len(facs)
        for f in prob.factors[L["first"]:L["first"] + L["count"]]:
            f = f.copy(); f["lm"] = k
            facs.append(f)
        lms.append((first, L["count"], L["anchor"], L["mask"]))
    facs = np.array(facs, ba_gen.FACTOR_DTYPE) if facs else prob.factors[:0]
    lms = np.array(lms, ba_gen.LM_DTYPE) if lms else prob.landmarks[:0]
    inv = prob.inv_depth[mine] if mine else prob.inv_depth[:0]
    first_rank = rank == 0
    return WindowProblem(prob.pose, prob.speed_bias, prob.ex_pose, prob.td[0], inv, facs, lms, prob.imu if first_rank else prob.imu[:0], prob.c.use_imu, prob.c.plane_kind,
                         prob.c.max_iters, prob.c.g_norm, prob.prior if first_rank else None, prob.prior_A if first_rank else None, prob.prior_b if first_rank else None)


def main(out):
    oracle = oracle_py.load()
    ctx = Context(width=64, height=64, max_cnt=10, min_dist=5)
    prob = ba_gen.make_window(oracle, seed=21, nframes=6, nlm=90, use_imu=1, with_prior=True)
    c, S, g = ba_eval(ctx, prob)
    res = {"full": np.concatenate([S.ravel(), g, [c]]), "n": np.array([len(g)])}
    for world in (2, 4):
        for r in range(world):
            cr, Sr, gr = ba_eval(ctx, share(prob, r, world))
            assert len(gr) == len(g)
            res[f"w{world}_r{r}"] = np.concatenate([Sr.ravel(), gr, [cr]])
    np.savez_compressed(out, **res)
    tot = sum(res[f"w2_r{r}"] for r in range(2))
    print("rel err of the 2-rank sum:", np.abs(tot - res["full"]).max() / np.abs(res["full"]).max())
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1])
