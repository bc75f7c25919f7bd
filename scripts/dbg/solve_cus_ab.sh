#!/bin/bash
# on the GPU box: DVINS_SOLVE_CUS=k (k CUs of every XCD reserved for the batched window solves) against the default, 16 and 64 sequences
cd $GRAFT_REPO_ROOT
run() { python bench.py --sequences $1 --batched --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; r=d.get('roofline') or {}; print('S=$1 SOLVE_CUS=${DVINS_SOLVE_CUS:-0}', d['value'], c.get('second_block_value'), 'bit_identity', (c.get('bit_identity') or {}).get('equal_to_single_thread_unbatched_run'), 'stages us', r.get('stages_us') or {k:v for k,v in r.items() if 'us' in k})"; }
for S in 16 64; do
  for sp in 0 1; do
    if [ $sp = 0 ]; then unset DVINS_EVAL_SPLIT; else export DVINS_EVAL_SPLIT=1; fi
    for k in $( [ $S = 16 ] && echo 0 2 4 || echo 0 4 8 ); do
      if [ $k = 0 ]; then unset DVINS_SOLVE_CUS; else export DVINS_SOLVE_CUS=$k; fi
      echo -n "EVAL_SPLIT=$sp "; run $S; echo -n "EVAL_SPLIT=$sp "; run $S
    done
  done
done
