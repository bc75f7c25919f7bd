#!/bin/bash
# on the GPU box: the default line (30 steps, raw only) with each abbuild/ library, interleaved twice; prints be_solve's HIP-event time and the frame rate
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = default ]; then unset DVINS_HIP_LIB; else export DVINS_HIP_LIB=$PWD/abbuild/libdvins_$lib.so; fi
  python3 bench.py --steps 40 --no-extra-lines --no-dynamic-line --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value', d['value'], 'blocks', d['config']['block_values'], 'be_solve_us', d['roofline']['kernels_us'].get('be_solve'), 'iters', d['config']['solver_iterations_per_frame'], 'ate', d['config']['ate_rmse_m_vs_ground_truth'])"
done
done
