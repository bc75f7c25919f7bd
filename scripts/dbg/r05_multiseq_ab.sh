#!/bin/bash
# on the GPU box: the multi-sequence line with each abbuild/ library given ("default" = the shipped one), S = 16 and 64
cd ${GRAFT_REPO_ROOT:-.}
export BENCH_BIT_IDENTITY=0
for S in 16 64; do for lib in "$@"; do
  if [ "$lib" = default ]; then unset DVINS_HIP_LIB; else export DVINS_HIP_LIB=$PWD/abbuild/libdvins_$lib.so; fi
  python3 bench.py --sequences $S --batched --steps 40 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
st=(d.get('roofline') or {}).get('stages') or {}
print('S=$S $lib', None if d is None else (d['value'], d['config']['second_block_value'], d['config']['ate_rmse_m_vs_ground_truth_max'], {k: v['avg_launch_us'] for k, v in st.items()}))"
done; done
