# second sweep: group sizes for 16 sequences, threads for 32 sequences (default groups of 8); then the BA-only micro-benchmark and two more long runs
cd /root/repo
export GPU_MAX_HW_QUEUES=12
run() { tag="$1"; shift; timeout 300 python bench.py "$@" --batched --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']; print('AB $tag', d['value'], c.get('second_block_value'), c.get('group_size'), c.get('runner_threads'), c.get('bit_identity',{}).get('equal_to_single_thread_unbatched_run'))
" | tee -a gpurun_out/ab_threads2.txt; }
for rep in 1 2; do
run s16_g4_t8 --sequences 16
run s16_g8_t8 --sequences 16 --group-size 8 --runner-threads 8
run s16_g16_t8 --sequences 16 --group-size 16 --runner-threads 8
run s32_g8_t8 --sequences 32
run s32_g8_t16 --sequences 32 --runner-threads 16
run s32_g8_t32 --sequences 32 --runner-threads 32
run s32_g16_t16 --sequences 32 --group-size 16 --runner-threads 16
done
timeout 600 python tests/tools/ba_microbench.py > gpurun_out/r06_ba_microbench.json 2> gpurun_out/ba_microbench.err; tail -2 gpurun_out/ba_microbench.err
timeout 900 python scripts/longrun_parity.py dynamic_static 500 640 360 > gpurun_out/r06_longrun_dynamic_static_640x360_500.json 2> gpurun_out/lr1.err; tail -2 gpurun_out/lr1.err
timeout 1500 python scripts/longrun_parity.py dynamic 600 1280 720 > gpurun_out/r06_longrun_dynamic_1280x720_600.json 2> gpurun_out/lr2.err; tail -2 gpurun_out/lr2.err
