cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_6_$tag.json 2> gpurun_out/r04_6_$tag.err; rc=$?; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_6_$tag.json").read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
    print("$tag rc=$rc", d["value"], c["second_block_value"], "groups of", c["group_size"], "threads", c["runner_threads"], "max ATE", c["ate_rmse_m_vs_ground_truth_max"], {k: v["avg_launch_us"] for k, v in (r.get("stages") or {}).items()})
except Exception as e: print("$tag rc=$rc NO LINE", open("gpurun_out/r04_6_$tag.err").read()[-200:])
PY
}
run s16_g4_a --sequences 16 --batched --group-size 4 --steps 60
run s16_g8_a --sequences 16 --batched --group-size 8 --steps 60
run s16_g4_b --sequences 16 --batched --group-size 4 --steps 60
run s16_g8_b --sequences 16 --batched --group-size 8 --steps 60
run s16_g2   --sequences 16 --batched --group-size 2 --steps 60
run s16_unbatched_t4 --sequences 16 --runner-threads 4 --steps 60
