cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_6_$tag.json 2>gpurun_out/r04_6_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_6_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$tag", d["value"], d["config"].get("second_block_value"), {k:v["avg_launch_us"] for k,v in (r.get("stages") or {}).items()})
except Exception as e: print("$tag ERR", e)
PY
}
run s16_g2_t8 --sequences 16 --batched --group-size 2 --runner-threads 8 --steps 40
run s16_g4_t4 --sequences 16 --batched --group-size 4 --runner-threads 4 --steps 40
run s16_g4_t2 --sequences 16 --batched --group-size 4 --runner-threads 2 --steps 40
run s24_g4_t6 --sequences 24 --batched --group-size 4 --runner-threads 6 --steps 40
run s32_g4_t8 --sequences 32 --batched --group-size 4 --runner-threads 8 --steps 40
run s32_g8_t4 --sequences 32 --batched --group-size 8 --runner-threads 4 --steps 40
run s16_nobatch_t4 --sequences 16 --runner-threads 4 --steps 40
run s8_g4_t2 --sequences 8 --batched --group-size 4 --runner-threads 2 --steps 40
python bench.py --steps 100 --no-cpu-baseline --no-dynamic-line > gpurun_out/r04_6_default.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04_6_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['config'].get('block_values'), d['roofline']['kernels_us'])"
