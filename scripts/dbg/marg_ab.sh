#!/bin/bash
# on the GPU box: be_marg_finish on the matrix cores (default) against the 4-wide panel form (DVINS_MARG_GENERIC=1): HIP-event time of the three marginalization launches, frame rate, and rocprofv3 averages
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for g in 0 1; do
  if [ $g = 1 ]; then export DVINS_MARG_GENERIC=1; else unset DVINS_MARG_GENERIC; fi
  python3 bench.py --steps 40 --no-extra-lines --no-dynamic-line --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('generic=$g', 'value', d['value'], 'blocks', d['config']['block_values'], 'be_marg_us(lm+sum+finish)', d['roofline']['kernels_us'].get('be_marg'), 'ate', d['config']['ate_rmse_m_vs_ground_truth'], 'iters', d['config']['solver_iterations_per_frame'])"
done; done
unset DVINS_MARG_GENERIC
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
  if [ $g = 1 ]; then export DVINS_MARG_GENERIC=1; else unset DVINS_MARG_GENERIC; fi
  rm -rf /tmp/marg_prof_$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/marg_prof_$g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > /dev/null 2>&1
  f=$(find /tmp/marg_prof_$g -name "*kernel_stats.csv" | head -1)
  echo "generic=$g rocprofv3:"; grep "be_marg\|be_solve_kernel" $f | cut -d, -f1-5 | cut -c1-160
done
