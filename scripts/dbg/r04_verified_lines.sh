# multi-sequence lines on the verified path (one host thread per group), every one behind the ATE gate; and the team regression test
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" > gpurun_out/r04_5_$tag.json 2> gpurun_out/r04_5_$tag.err; rc=$?; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_5_$tag.json").read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
    print("$tag rc=$rc", d["value"], c["second_block_value"], "threads/group", c["host_threads_per_group"], "teams", c["teams_experimental"], "max ATE", c["ate_rmse_m_vs_ground_truth_max"], {k: v["avg_launch_us"] for k, v in (r.get("stages") or {}).items()})
except Exception as e: print("$tag rc=$rc NO LINE", open("gpurun_out/r04_5_$tag.err").read()[-300:])
PY
}
run s16 --sequences 16 --batched --steps 60
run s16_again --sequences 16 --batched --steps 60
run s16_g16 --sequences 16 --batched --group-size 16 --steps 60
run s32_g16 --sequences 32 --batched --group-size 16 --steps 40
run s32_g8 --sequences 32 --batched --group-size 8 --steps 40
run s64_g16 --sequences 64 --batched --group-size 16 --steps 40
run k21 --config kitti --sequences 21 --batched --steps 60
run k21_g7 --config kitti --sequences 21 --batched --group-size 7 --steps 60
run s16_teams_gate_demo --sequences 16 --batched --teams --runner-threads 4 --steps 60
python -m pytest tests/test_runner.py tests/test_batch.py -q -m gpu -rx 2>&1 | tail -8
