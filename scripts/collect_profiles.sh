#!/bin/bash
# On the GPU box (gpurun): the rocprofv3 summaries profiles/ keeps for a round.  usage: scripts/collect_profiles.sh <tag>   (e.g. r03_b) -> gpurun_out/<tag>_*
# --kernel-trace --stats for the default and the dynamic workload, and the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only — the
# pool refuses --pmc combined with other trace domains) that profiles/pmc_traffic.json is made of.
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_dyn -- python3 $ROOT/bench.py --mode dynamic --steps 30 --no-cpu-baseline > $OUT/${TAG}_bench_dynamic_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
BATCHED="--sequences 16 --batched --group-size 8 --runner-threads 2 --steps 30"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_bat -- python3 $ROOT/bench.py $BATCHED > $OUT/${TAG}_bench_batched_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
cd $ROOT
# the --stats table of OUR kernels (the PyTorch kernels of the synthetic-image renderer filtered out)
for pair in "trace bench" "trace_dyn bench_dynamic" "trace_bat bench_batched"; do set -- $pair; f=$(ls $OUT/${TAG}_$1/*/*kernel_stats.csv | head -1); (head -1 $f; grep -v "at::\|Cijk\|elementwise\|rocclr\|^\"Name" $f) > $OUT/${TAG}_$2_kernel_stats.csv; done
python3 scripts/pmc_to_json.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.json
python3 scripts/pmc_to_json.py $OUT/${TAG}_pmc_fetch_bat $OUT/${TAG}_pmc_write_bat --batched 8 > $OUT/${TAG}_pmc_traffic_batched.json
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.txt || true
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_trace_dyn $OUT/${TAG}_trace_bat $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_fetch_bat $OUT/${TAG}_pmc_write_bat
ls -la $OUT | grep ${TAG}
