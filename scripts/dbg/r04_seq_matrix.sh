cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_3_$tag.json 2>gpurun_out/r04_3_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_3_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$tag", d["value"], d["config"].get("second_block_value"), {k:v["avg_launch_us"] for k,v in (r.get("stages") or {}).items()}, d["config"].get("front_end_launches"))
except Exception as e: print("$tag ERR", e)
PY
}
run s16_g8_t2 --sequences 16 --batched --group-size 8 --runner-threads 2 --steps 40
run s16_g8_t2_nofront --sequences 16 --batched --group-size 8 --runner-threads 2 --steps 40 --no-batch-front
run s16_g16_t1 --sequences 16 --batched --group-size 16 --runner-threads 1 --steps 40
run s16_g4_t4 --sequences 16 --batched --group-size 4 --runner-threads 4 --steps 40
run s16_g8_t1 --sequences 16 --batched --group-size 8 --runner-threads 1 --steps 40
run s21k_g21_t1 --config kitti --sequences 21 --batched --group-size 21 --runner-threads 1 --steps 40
run s21k_g7_t3 --config kitti --sequences 21 --batched --group-size 7 --runner-threads 3 --steps 40
nproc
