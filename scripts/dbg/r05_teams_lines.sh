#!/bin/bash
# on the GPU box: multi-sequence lines, teams of two (default) / of four / one thread per group
cd ${GRAFT_REPO_ROOT:-.}
for S in 16 32 64; do for cfg in "" "--runner-threads 16" "--no-teams"; do
  python3 bench.py --sequences $S --batched --steps 40 $cfg 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('S=$S [$cfg]', None if d is None else (d['value'], d['config']['second_block_value'], d['config']['runner_threads'], d['config']['group_size'], d['config']['ate_rmse_m_vs_ground_truth_max'], (d['config'].get('bit_identity') or {}).get('equal_to_single_thread_unbatched_run')))"
done; done
