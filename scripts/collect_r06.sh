#!/bin/bash
# On the GPU box (gpurun), LAST step of a round: everything profiles/ keeps, collected at the sources that ship.  usage: scripts/collect_r06.sh <tag>   (-> gpurun_out/<tag>_*)
#   1. rocprofv3 --kernel-trace --stats: default, dynamic and batched (32 sequences in groups of 16) workloads -> <tag>_bench*_kernel_stats.csv
#   2. the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of the default and of the batched command -> <tag>_pmc_traffic*.json; the JSON
#      carries the digest of the kernel sources (bench.csrc_digest): bench.py reports roofline.traffic only while it matches
#   3. the bench lines (no profiler): default (with dynamic_line, host_frames_line, multiseq_line inside), --steps 20 --warmup 5 (the driver's command), dynamic,
#      16 / 32 / 64 sequences, 21 KITTI-size sequences, 16 without teams — each multi-sequence line behind the ATE gate and the two-member bit-identity check
#   4. (round 6) the worst frame of a cold dynamic sequence, three runs + one with DVINS_COPY_ENGINE=1
# The script REFUSES to finish (exit 4) if the digest in the PMC file it wrote differs from the sources' digest, or if the default line's roofline.traffic is null.
set -e
trap 'echo "collect: failed at line $LINENO"' ERR
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8          # (under rocprofv3 the runtime is initialised before bench.py can set it; the dynamic line needs 8 — exported for every profiled command alike)
BATCHED="--sequences 32 --batched --group-size 16 --steps 30"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/${TAG}_*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_trace -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_trace_dyn -- python3 $ROOT/bench.py --mode dynamic --steps 30 --no-cpu-baseline > $OUT/${TAG}_bench_dynamic_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc_fetch -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc_write -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > /dev/null 2>&1
export GPU_MAX_HW_QUEUES=12
export BENCH_BIT_IDENTITY=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_trace_bat -- python3 $ROOT/bench.py $BATCHED > $OUT/${TAG}_bench_batched_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc_fetch_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/${TAG}_pmc_write_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
unset BENCH_BIT_IDENTITY GPU_MAX_HW_QUEUES
cd $ROOT
for pair in "trace bench" "trace_dyn bench_dynamic" "trace_bat bench_batched"; do set -- $pair; f=$(find /tmp/${TAG}_$1 -name "*kernel_stats.csv" | head -1); (head -1 $f; grep -v "at::\|Cijk\|elementwise\|rocclr\|^\"Name" $f) > $OUT/${TAG}_$2_kernel_stats.csv; done
python3 scripts/pmc_to_json.py /tmp/${TAG}_pmc_fetch /tmp/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.json
python3 scripts/pmc_to_json.py /tmp/${TAG}_pmc_fetch_bat /tmp/${TAG}_pmc_write_bat --batched 16 > $OUT/${TAG}_pmc_traffic_batched.json
python3 scripts/pmc_summary.py /tmp/${TAG}_pmc_fetch /tmp/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.txt || true
# the bench reads profiles/pmc_traffic*.json: put the fresh ones there BEFORE the bench lines, so that the lines carry roofline.traffic
cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json; cp $OUT/${TAG}_pmc_traffic_batched.json profiles/pmc_traffic_batched.json
run() { name=$1; shift; python3 bench.py "$@" > $OUT/${TAG}_bench${name}.json 2> $OUT/${TAG}_bench${name}.err || echo "bench${name}: exit $? (no line: see ${TAG}_bench${name}.err)"; }
run ""
run _steps20                --steps 20 --warmup 5
run _dynamic                --mode dynamic --no-cpu-baseline
run _sequences16_batched    --sequences 16 --batched --steps 60
run _sequences16_no_teams   --sequences 16 --batched --no-teams --steps 60
run _sequences32_batched    --sequences 32 --batched --steps 40
run _sequences64_batched    --sequences 64 --batched --steps 40
run _kitti21_batched        --config kitti --sequences 21 --batched --steps 60
run _sequences8             --sequences 8 --runner-threads 2 --steps 40
run _steps20_b              --steps 20 --warmup 5
# worst frame of a dynamic 1280x720 sequence from a cold process (scripts/dyn_cold_frames.py; tests/test_frame_gaps.py), copy kernels vs the copy engines
for i in 1 2 3; do python3 scripts/dyn_cold_frames.py 60 0 | tail -1 > $OUT/${TAG}_cold_dynamic_frames_$i.json; done
DVINS_COPY_ENGINE=1 python3 scripts/dyn_cold_frames.py 60 0 | tail -1 > $OUT/${TAG}_cold_dynamic_frames_copy_engine.json
for f in $OUT/${TAG}_bench*.err; do [ "$(grep -v amdgpu.ids $f | wc -c)" -le 1 ] && rm -f $f; done
python3 - <<PY
import json, sys
sys.path.insert(0, "$ROOT")
from bench import csrc_digest
d = csrc_digest()
for name in ("pmc_traffic.json", "pmc_traffic_batched.json"):
    got = json.load(open("$OUT/${TAG}_" + name)).get("csrc_digest")
    if got != d:
        print("collect: digest of", name, got, "differs from the sources", d); sys.exit(4)
line = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
r = line.get("roofline") or {}
print("collect: default line", line["value"], "roofline.traffic", r.get("traffic"), "stale", r.get("traffic_stale"), "| dynamic", (line["config"].get("dynamic_line") or {}).get("value"),
      "| host frames", (line["config"].get("host_frames_line") or {}).get("value"), "| 16 sequences", (line["config"].get("multiseq_line") or {}).get("value"))
if r.get("traffic") is None or r.get("traffic_stale"):
    print("collect: the default line carries no roofline.traffic"); sys.exit(4)
PY
ls $OUT | grep ${TAG}_
