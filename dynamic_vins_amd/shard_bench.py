"""bench.py --shard: ONE window solve sharded by landmark over the ranks; see dist.py.  Filled in with the sharded solve."""


def run_shard_bench(args, rank, world, local_rank):
    raise SystemExit("bench.py --shard: not available in this build")
