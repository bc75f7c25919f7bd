cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_4_$tag.json 2>gpurun_out/r04_4_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04_4_$tag.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$tag", d["value"], d["config"].get("second_block_value"), {k:(v["avg_launch_us"], v.get("frac")) for k,v in (r.get("stages") or {}).items()})
except Exception as e: print("$tag ERR", e)
PY
}
run s32_g16_t8 --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 30
run s48_g16_t12 --sequences 48 --batched --group-size 16 --runner-threads 12 --steps 30
run s64_g16_t16 --sequences 64 --batched --group-size 16 --runner-threads 16 --steps 30
run s64_g32_t16 --sequences 64 --batched --group-size 32 --runner-threads 16 --steps 30
run s64_g16_t8 --sequences 64 --batched --group-size 16 --runner-threads 8 --steps 30
run s42k_g21_t14 --config kitti --sequences 42 --batched --group-size 21 --runner-threads 14 --steps 30
