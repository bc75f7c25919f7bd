set -x
cd /root/repo
export GPU_MAX_HW_QUEUES=12
timeout 900 python -m pytest tests/test_batch.py tests/test_runner.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in 0 1; do
  for n in 16 64; do
    DVINS_BATCH_UPLOAD=$v timeout 300 python bench.py --sequences $n --batched --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('AB upload=$v n=$n', d['value'], d['config'].get('block_values'))
" | tee -a gpurun_out/ab_upload.txt
  done
done
done
