cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_10_$tag.json 2>gpurun_out/r04_10_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_10_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$tag", d["value"], d["config"].get("second_block_value"), {k:(v["avg_launch_us"], v["achieved_GBs"]) for k,v in (r.get("stages") or {}).items()})
except Exception as e: print("$tag ERR", e, open("gpurun_out/r04_10_$tag.err").read()[-300:])
PY
}
run s16_g16_t2 --sequences 16 --batched --group-size 16 --runner-threads 2 --steps 40
run s16_g16_t4 --sequences 16 --batched --group-size 16 --runner-threads 4 --steps 40
run s16_g16_t8 --sequences 16 --batched --group-size 16 --runner-threads 8 --steps 40
run s16_g8_t4 --sequences 16 --batched --group-size 8 --runner-threads 4 --steps 40
run s16_g8_t8 --sequences 16 --batched --group-size 8 --runner-threads 8 --steps 40
run s32_g16_t8 --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 40
run s32_g16_t4 --sequences 32 --batched --group-size 16 --runner-threads 4 --steps 40
run s21k_g21_t3 --config kitti --sequences 21 --batched --group-size 21 --runner-threads 3 --steps 40
run s21k_g21_t7 --config kitti --sequences 21 --batched --group-size 21 --runner-threads 7 --steps 40
