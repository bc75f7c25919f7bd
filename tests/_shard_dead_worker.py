"""worker of tests/test_sharded_solve.py::test_peer_transport_reports_a_silent_peer (GPU box, two processes on the one GPU):
both ranks set up the one-shot peer transport; rank 1 then NEVER takes part in an exchange (a dead or hung peer as rank 0 sees it) and only waits for rank 0
at the final barrier.  Rank 0 solves a sharded window: its wait kernel must give up after the configured time, the solve must return an ERROR (not a
result built from a stale buffer), later exchanges must not wait again, and dv_dist_info must report the same error."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from dynamic_vins_amd import _abi, dist as dv_dist
from dynamic_vins_amd.backend import ba_solve
from dynamic_vins_amd.frontend import Context
from tests import ba_gen, oracle_py

rank, world, _ = dv_dist.init(prefer_gpu=False)
shard = Context(width=64, height=64, max_cnt=10, min_dist=5)
dv_dist.shard_window(shard, rank, world, transport="peer")
assert shard.lib.dv_debug_set(shard.h, b"peer_timeout_ms", 300) == 0
dv_dist.barrier()
if rank == 0:
    oracle = oracle_py.load()
    w = ba_gen.make_window(oracle, seed=2, with_prior=True)
    t0 = time.perf_counter()
    try:
        ba_solve(shard, w.clone())
        first = "NO ERROR"
    except _abi.DvinsError as e:
        first = str(e)
    t1 = time.perf_counter()
    try:
        ba_solve(shard, w.clone())
        second = "NO ERROR"
    except _abi.DvinsError as e:
        second = str(e)
    t2 = time.perf_counter()
    try:
        dv_dist.dist_info(shard)
        info = "NO ERROR"
    except _abi.DvinsError as e:
        info = str(e)
    with open(os.path.join(sys.argv[1], "dead_rank0.txt"), "w") as f:
        f.write("%s\n%s\n%s\n%.3f %.3f\n" % (first, second, info, t1 - t0, t2 - t1))
dv_dist.barrier()
shard.close()
dv_dist.finalize()
