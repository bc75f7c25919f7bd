"""worker of tests/test_distributed.py: run as  python -m torch.distributed.run --nproc-per-node 2 tests/_dist_worker.py OUT"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from dynamic_vins_amd import dist as dv_dist

rank, world, local_rank = dv_dist.init(prefer_gpu=False)
dv_dist.barrier()
res = {"rank": rank, "world": world}
# timing protocol of bench.py: MAX over ranks
res["tmax"] = dv_dist.max_over_ranks(1.0 + rank)
res["sum"] = dv_dist.sum_over_ranks(rank + 1)
# landmark sharding is a partition
nlm = 301
mine = dv_dist.shard_landmarks(nlm, rank, world)
res["mine"] = len(mine)
# the exchange step: partial reduced systems of a synthetic window (values chosen to be order sensitive in fp64)
n = 165
rng = np.random.default_rng(1234)
w = rng.normal(0, 1, (nlm, n)) * np.exp(rng.normal(0, 6, (nlm, 1)))        # wide dynamic range
gl = rng.normal(0, 1, nlm)
S = np.zeros((n, n)); g = np.zeros(n); cost = 0.0
for l in mine:
    S += np.outer(w[l], w[l]); g += w[l] * gl[l]; cost += 0.5 * gl[l] ** 2
part = torch.from_numpy(np.concatenate([S.ravel(), g, [cost]]))
tot = dv_dist.allreduce_reduced_system(part)
# the same sum formed locally in rank order
ref = None
for r in range(world):
    Sr = np.zeros((n, n)); gr = np.zeros(n); cr = 0.0
    for l in dv_dist.shard_landmarks(nlm, r, world):
        Sr += np.outer(w[l], w[l]); gr += w[l] * gl[l]; cr += 0.5 * gl[l] ** 2
    v = np.concatenate([Sr.ravel(), gr, [cr]])
    ref = v if ref is None else ref + v
res["bitwise_rank_ordered"] = bool(np.array_equal(tot.numpy(), ref))
res["digest"] = float(np.abs(tot.numpy()).sum())
full = sum(np.outer(w[l], w[l]) for l in range(nlm))
res["rel_err_vs_unsharded"] = float(np.abs(tot.numpy()[: n * n].reshape(n, n) - full).max() / np.abs(full).max())
res["rate"] = dv_dist.whole_job_rate(100, world, res["tmax"])
dv_dist.barrier()
with open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w") as f:
    json.dump(res, f)
dv_dist.finalize()
