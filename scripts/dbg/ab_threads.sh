# sweep of host threads per dv_batch group / group counts for the multi-sequence lines, one box
cd /root/repo
export GPU_MAX_HW_QUEUES=12
run() { tag="$1"; shift; timeout 300 python bench.py "$@" --batched --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']; print('AB $tag', d['value'], c.get('second_block_value'), c.get('group_size'), c.get('runner_threads'), c.get('bit_identity',{}).get('equal_to_single_thread_unbatched_run'))
" | tee -a gpurun_out/ab_threads.txt; }
for rep in 1 2; do
run s16_t8 --sequences 16
run s16_t16 --sequences 16 --runner-threads 16
run s16_t12 --sequences 16 --runner-threads 12
run s64_t8 --sequences 64
run s64_t16 --sequences 64 --runner-threads 16
run s64_t32 --sequences 64 --runner-threads 32
run s64_g8_t16 --sequences 64 --group-size 8 --runner-threads 16
run s64_g8_t32 --sequences 64 --group-size 8 --runner-threads 32
done
# second sweep (another box): group sizes for 16 sequences, threads for 32 sequences
# run s16_g4_t8 --sequences 16
# run s16_g8_t8 --sequences 16 --group-size 8 --runner-threads 8
# run s16_g16_t8 --sequences 16 --group-size 16 --runner-threads 8
# run s32_g8_t8 --sequences 32
# run s32_g8_t16 --sequences 32 --runner-threads 16
# run s32_g8_t32 --sequences 32 --runner-threads 32
# run s32_g16_t16 --sequences 32 --group-size 16 --runner-threads 16
