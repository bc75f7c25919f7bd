#!/bin/bash
# on the GPU box: groupings x team sizes for the multi-sequence lines (teams default)
cd ${GRAFT_REPO_ROOT:-.}
export BENCH_BIT_IDENTITY=0
run() { python3 bench.py --sequences $1 --batched --steps 40 --group-size $2 --runner-threads $3 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('S=$1 group_size=$2 threads=$3', None if d is None else (d['value'], d['config']['second_block_value'], d['config']['ate_rmse_m_vs_ground_truth_max']))"; }
run 16 8 8; run 16 8 16; run 16 16 8; run 16 16 16; run 16 4 12
run 32 16 8; run 32 16 16; run 32 8 12
run 64 32 8; run 64 32 16; run 64 16 12; run 64 16 24
