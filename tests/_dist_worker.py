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
# the exchange on RECORDED partials of the HIP path (tests/golden/shard_partials.npz, written on the GPU box by tests/golden/gen_shard_partials.py): each rank
# contributes the partial [S | g | cost] the kernels produced for its share of the landmarks; the rank-ordered sum must reproduce the unsharded system
gold = np.load(os.path.join(ROOT, "tests", "golden", "shard_partials.npz"))
mine_v = torch.from_numpy(gold[f"w{world}_r{rank}"].copy())
tot_v = dv_dist.allreduce_reduced_system(mine_v).numpy()
full = gold["full"]
res["gold_rel_err"] = float(np.abs(tot_v - full).max() / np.abs(full).max())
ordered = gold[f"w{world}_r0"].copy()
for r in range(1, world):
    ordered = ordered + gold[f"w{world}_r{r}"]
res["gold_bitwise_rank_ordered"] = bool(np.array_equal(tot_v, ordered))
res["gold_digest"] = float(np.abs(tot_v).sum())
nn = int(gold["n"][0])
Sg = tot_v[: nn * nn].reshape(nn, nn)
res["gold_symmetric"] = bool(np.abs(Sg - Sg.T).max() <= 1e-12 * np.abs(Sg).max())
dv_dist.barrier()
with open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w") as f:
    json.dump(res, f)
dv_dist.finalize()
