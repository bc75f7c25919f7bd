# kernel-trace timeline of the dynamic line: per-kernel averages + one steady-state frame launch by launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/dynprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/dynprof -- python3 $R/bench.py --mode dynamic --steps 60 --no-cpu-baseline > $R/gpurun_out/dyn_tl_bench.json 2> $R/gpurun_out/dyn_tl_bench.err
python3 $R/scripts/dbg/round_timeline.py /tmp/dynprof xp_detect -8 > $R/gpurun_out/${1:-dyn_timeline}.txt 2>&1
tail -c 600 $R/gpurun_out/dyn_tl_bench.json | head -c 300
