# stream priorities: back-end streams highest (1) / front-end streams lowest (2) against all default (0); one box, interleaved
cd /root/repo
one() { tag="$1"; shift; timeout 400 python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']; dl=c.get('dynamic_line') or {}
        print('AB $tag prio=$DVINS_BA_PRIORITY', d['value'], c.get('second_block_value'), 'dyn', dl.get('value'), dl.get('second_block_value'), (dl.get('block_step_ms') or [{}])[0].get('max'))
" | tee -a gpurun_out/ab_priority.txt; }
for rep in 1 2; do
for p in 0 1 2; do
  export DVINS_BA_PRIORITY=$p
  one single --no-extra-lines
  GPU_MAX_HW_QUEUES=12 one s16 --sequences 16 --batched
  GPU_MAX_HW_QUEUES=12 one s64 --sequences 64 --batched
done
done
