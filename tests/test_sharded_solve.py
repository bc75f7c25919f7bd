"""The landmark-sharded window solve (include/dvins.h dv_dist_*), `-m gpu`.

(1) one process, world 1, through every transport (RCCL all-gather on the BA stream; host call-back; one-shot peer writes): the sharded kernels (owned-range evaluation and
    reduction, exchange vector, rank-ordered finalize, cost exchange) against the unsharded solve — same iteration sequence, states within 1e-12;
(2) two processes on the one GPU (torch.distributed.run, gloo; host transport, and the one-shot peer transport whose windows the two processes map into
    each other through hipIpc): every rank returns the SAME bits, and the sharded solve reproduces the
    unsharded one to 1e-9 on operator-level windows (with / without prior, VO, 1 / 0 / 300 / 1000 landmarks) and through the estimator's fused path
    (solve + gauge fix + marginalization, device-resident prior) over 30 frames.
RCCL refuses two ranks on one device, so the RCCL transport with world > 1 is exercised by bench.py --shard on a multi-GPU node only."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import ba_gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("transport", ["rccl", "host", "peer"])
def test_world_one_sharded_path_matches_plain(gpu_ctx_factory, oracle, transport):
    from dynamic_vins_amd import dist as dv_dist
    from dynamic_vins_amd.backend import ba_eval, ba_solve
    shard = gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5)
    plain = gpu_ctx_factory(width=64, height=64, max_cnt=10, min_dist=5)
    dv_dist.shard_window(shard, 0, 1, transport=transport)
    info0 = dv_dist.dist_info(shard)
    assert info0["transport"] == transport and info0["rccl_ranks"] == (1 if transport == "rccl" else 0)      # ncclCommCount of the communicator
    for kw in [dict(seed=2, with_prior=True), dict(seed=6, nlm=300, max_iters=10, with_prior=True), dict(seed=3, use_imu=0, nframes=7), dict(seed=12, nlm=0, max_iters=4, with_prior=True)]:
        ref = ba_gen.make_window(oracle, **kw)
        a, b = ref.clone(), ref.clone()
        ca, Sa, ga = ba_eval(shard, a)
        cb, Sb, gb = ba_eval(plain, b)
        assert np.abs(Sa - Sb).max() <= 1e-13 * np.abs(Sb).max() and np.abs(ga - gb).max() <= 1e-13 * max(np.abs(gb).max(), 1e-300) and np.isclose(ca, cb, rtol=1e-14)
        sa, sb = ba_solve(shard, a), ba_solve(plain, b)
        assert (sa.iterations, sa.termination, sa.successful) == (sb.iterations, sb.termination, sb.successful)
        assert np.abs(a.pose - b.pose).max() < 1e-12 and np.abs(a.inv_depth - b.inv_depth).max(initial=0.0) < 1e-12
        assert np.isclose(sa.final_cost, sb.final_cost, rtol=1e-12)
    assert dv_dist.dist_info(shard)["exchanges"] > 0
    # round 5: the per-linearisation exchange does not grow with the window (<= 110 KB at L = 1000) — partial reduced system + quadratic-form coefficients
    for L in (0, 254, 1000):
        eb = dv_dist.exchange_bytes(shard, L)
        assert eb["system"] == 12424 * 8 <= 110 * 1024 and eb["cost"] == 64 and eb["depth"] == 8 * max(1, L)
    assert shard.lib.dv_dist_shutdown(shard.h) == 0
    # after shutdown the ctx solves unsharded again
    ref = ba_gen.make_window(oracle, seed=2, with_prior=True)
    a, b = ref.clone(), ref.clone()
    ba_solve(shard, a); ba_solve(plain, b)
    assert np.array_equal(a.pose, b.pose)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_two_ranks_on_one_gpu_reproduce_the_unsharded_solve(tmp_path, transport):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_shard_worker.py"), str(tmp_path), transport]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    R = [np.load(tmp_path / f"shard_rank{k}.npz") for k in range(2)]
    assert R[0]["info"][1] == 2 and R[1]["info"][0] == 1 and R[0]["info"][2] > 0 and R[0]["info"][2] == R[1]["info"][2]
    n_cases = len([k for k in R[0].files if k.startswith("solve") and k.endswith("_sum")])
    assert n_cases == 7
    for i in range(n_cases):
        # identical bits on both ranks (rank-ordered sums)
        assert np.array_equal(R[0][f"solve{i}_shard"], R[1][f"solve{i}_shard"]), i
        assert np.array_equal(R[0][f"eval{i}_digest"], R[1][f"eval{i}_digest"]), i
        # the reduced camera system: sharded == unsharded up to summation order
        assert (R[0][f"eval{i}_S"] < 1e-12).all(), (i, R[0][f"eval{i}_S"])
        s = R[0][f"solve{i}_sum"]
        assert s[0] == s[1] and s[2] == s[3], (i, s)                   # iterations, termination
        assert np.isclose(s[4], s[5], rtol=1e-9) and np.isclose(s[6], s[7], rtol=1e-12)
        assert np.abs(R[0][f"solve{i}_shard"] - R[0][f"solve{i}_plain"]).max() < 1e-9, i
    # estimator: 30 frames through the fused path
    assert np.array_equal(R[0]["est_shard"], R[1]["est_shard"])
    its = R[0]["est_its"]
    assert np.array_equal(its[:, 0], its[:, 1]) and np.array_equal(its[:, 2], its[:, 3])
    assert np.abs(R[0]["est_shard"][:, :, :7] - R[0]["est_plain"][:, :, :7]).max() < 1e-9
    # operator: rank-ordered sum of v_r = arange * (1 + r) + 0.1 r
    want = np.arange(1000, dtype=np.float64) * 1.0 + (np.arange(1000, dtype=np.float64) * 2.0 + 0.1)
    assert np.array_equal(R[0]["allreduce"], want) and np.array_equal(R[1]["allreduce"], want)


def test_peer_transport_reports_a_silent_peer(tmp_path):
    """failure path of the one-shot peer transport: a peer that never raises its flag turns into an ERROR of the solve that needed it (bounded wait on the
    device, flag in pinned memory read at the collect), the wait is paid once (sticky), and dv_dist_info reports it — never a result from a stale buffer"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_shard_dead_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    first, second, info, times = open(tmp_path / "dead_rank0.txt").read().splitlines()
    for msg in (first, second, info):
        assert "peer did not deliver" in msg, (first, second, info)
    t_first, t_second = map(float, times.split())
    assert 0.25 < t_first < 30.0          # one time-out of 0.3 s (plus the slots enqueued behind it, which no longer wait)
    assert t_second < t_first             # sticky: a known-dead peer is not waited for again
