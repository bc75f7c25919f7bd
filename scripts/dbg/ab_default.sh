# A/B of the default bench line on ONE box: the round-3 tree (dbg_old/tree) against the working tree, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  (cd dbg_old/tree && python bench.py --steps 100 --no-cpu-baseline --no-dynamic-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['value'], d['config'].get('block_values'))")
  python bench.py --steps 100 --no-cpu-baseline --no-dynamic-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['config'].get('block_values'))"
done
