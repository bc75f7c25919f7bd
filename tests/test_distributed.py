"""world_size-2 test of the N > 1 path on CPU (gloo): launch protocol of bench.py (torch.distributed.run, env
rendezvous on 127.0.0.1), barrier + MAX-over-ranks timing, landmark sharding and the rank-ordered all-reduce of the
reduced camera system."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo(tmp_path):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["rank"] for x in res] == [0, 1] and all(x["world"] == 2 for x in res)
    assert all(x["tmax"] == 2.0 for x in res)              # MAX over ranks
    assert all(x["sum"] == 3.0 for x in res)
    assert res[0]["mine"] + res[1]["mine"] == 301          # the shards partition the landmarks
    assert all(x["bitwise_rank_ordered"] for x in res)     # rank-ordered sum, same bits on both ranks
    assert res[0]["digest"] == res[1]["digest"]
    assert all(x["rel_err_vs_unsharded"] < 1e-12 for x in res)
    assert all(x["rate"] == 100.0 for x in res)            # 2 ranks x 100 frames / 2.0 s
    # recorded HIP partials of a window sharded over 2 ranks: the exchange reproduces the unsharded reduced camera system
    assert all(x["gold_rel_err"] < 1e-13 for x in res), [x["gold_rel_err"] for x in res]
    assert all(x["gold_bitwise_rank_ordered"] and x["gold_symmetric"] for x in res)
    assert res[0]["gold_digest"] == res[1]["gold_digest"]


def test_single_process_helpers_are_identity():
    sys.path.insert(0, ROOT)
    import torch
    from dynamic_vins_amd import dist as d
    assert d.max_over_ranks(3.5) == 3.5 and d.sum_over_ranks(2.0) == 2.0
    t = torch.arange(5, dtype=torch.float64)
    assert torch.equal(d.allreduce_reduced_system(t), t)
    assert d.shard_landmarks(7, 0, 1) == list(range(7))
    assert d.shard_landmarks(7, 0, 2) == [0, 1, 2, 3] and d.shard_landmarks(7, 1, 2) == [4, 5, 6] and d.shard_landmarks(2, 3, 4) == []
    assert d.whole_job_rate(50, 1, 0.5) == 100.0


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher must not run as one rank labelled n_gpus 1: it starts its two ranks itself as fresh child processes
    (torch.distributed.run, rendezvous on 127.0.0.1) before anything touches the GPU.  Rehearsed here on CPU (gloo, BENCH_SELFTEST: rendezvous, all-gather
    of the rank ids, the barrier + MAX-over-ranks of the timed region; no compute)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(BENCH_SELFTEST="1", BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout                       # ONE line, from rank 0
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["requested_gpus"] == 2 and out["ranks"] == [0, 1] and out["local_ranks"] == [0, 1]
    assert out["distinct_processes"] == 2 and out["launcher"] is True and out["max_over_ranks"] == 2.0
    # under a launcher (the driver's form) the same command does not launch again
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env2, cwd=str(tmp_path))
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["launcher"] is False
