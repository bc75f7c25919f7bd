cd $GRAFT_REPO_ROOT
export BENCH_BIT_IDENTITY=0
run() { env $1 python3 bench.py --sequences 16 --batched --steps 60 $2 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print('$1 [$2]', None if d is None else (d['value'], d['config']['second_block_value']))"; }
for rep in 1 2; do
run A=1 ""
run A=1 "--runner-threads 12"
run GPU_MAX_HW_QUEUES=16 ""
run GPU_MAX_HW_QUEUES=10 ""
done
