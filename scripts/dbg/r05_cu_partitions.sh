#!/bin/bash
# on the GPU box: the multi-sequence lines with the groups' streams on disjoint CU partitions (DVINS_CU_PARTITIONS, dv_group_stream_create) against the default
cd ${GRAFT_REPO_ROOT:-.}
export BENCH_BIT_IDENTITY=0
for S in 16 64; do
for cfg in "0 0" "4 0" "4 1" "2 0"; do
  set -- $cfg
  if [ $1 = 0 ]; then unset DVINS_CU_PARTITIONS; else export DVINS_CU_PARTITIONS=$1; fi
  if [ $2 = 0 ]; then unset DVINS_CU_PARTITION_FRONT; else export DVINS_CU_PARTITION_FRONT=1; fi
  python3 bench.py --sequences $S --batched --steps 40 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('S=$S partitions=$1 front=$2', None if d is None else (d['value'], d['config']['second_block_value'], d['config']['ate_rmse_m_vs_ground_truth_max'], (d.get('roofline') or {}).get('stages')))"
done; done
