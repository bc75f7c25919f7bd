# how far does one GPU go?  128 sequences (33 GB of resident frames) in two layouts; each line behind the ATE gate and the bit-identity check like the others
cd /root/repo
export GPU_MAX_HW_QUEUES=12
run() { tag="$1"; shift; timeout 900 python bench.py "$@" --batched --no-cpu-baseline 2>gpurun_out/many_$tag.err | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']; print('MANY $tag', d['value'], c.get('second_block_value'), c.get('group_size'), c.get('runner_threads'), c.get('bit_identity',{}).get('equal_to_single_thread_unbatched_run'), c.get('ate_rmse_m_vs_ground_truth_max'))
" | tee -a gpurun_out/many_sequences.txt; tail -2 gpurun_out/many_$tag.err | cut -c1-300; }
run s128_g16_t64 --sequences 128 --group-size 16 --runner-threads 64 --steps 40
run s128_default --sequences 128 --steps 40
run s96_g16_t48 --sequences 96 --group-size 16 --runner-threads 48 --steps 40
run s256_default --sequences 256 --steps 40
run s128_g32_t64 --sequences 128 --runner-threads 64 --steps 40
