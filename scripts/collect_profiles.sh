#!/bin/bash
# On the GPU box (gpurun): everything profiles/ keeps for a round.  usage: scripts/collect_profiles.sh <tag>   (e.g. r04_a) -> gpurun_out/<tag>_*
#   * the bench lines themselves (no profiler): default (raw + dynamic line + cpu_baseline), --steps 20 (the driver's command), --mode dynamic, and the multi-sequence
#     lines of BASELINE.json's config 4, each behind the per-sequence ATE gate (a run with a corrupted trajectory prints no line): 16 x 1280x720 (four groups of 4, one host
#     thread per group: the default of --batched), 32 x 1280x720 in groups of 8 and of 16, 64 in groups of 16, 21 KITTI-size sequences in groups of 11 and of 7
#   * rocprofv3 --kernel-trace --stats for the default, the dynamic and the batched (32 sequences, groups of 16) workload
#   * the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only — the pool refuses --pmc combined with other trace domains) of the default and
#     of the batched command -> pmc_traffic.json / pmc_traffic_batched.json
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
BATCHED="--sequences 32 --batched --group-size 16 --steps 30"
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_steps20.json 2>/dev/null
python3 bench.py --mode dynamic --no-cpu-baseline > $OUT/${TAG}_bench_dynamic.json 2>/dev/null
python3 bench.py --sequences 16 --batched --steps 60 > $OUT/${TAG}_bench_sequences16_batched.json 2>$OUT/${TAG}_bench_sequences16_batched.err || true
python3 bench.py --sequences 16 --batched --teams --runner-threads 4 --steps 60 > $OUT/${TAG}_bench_sequences16_teams.json 2>/dev/null || true
python3 bench.py --sequences 32 --batched --group-size 8 --steps 40 > $OUT/${TAG}_bench_sequences32_groups8.json 2>/dev/null || true
python3 bench.py --sequences 32 --batched --group-size 16 --steps 40 > $OUT/${TAG}_bench_sequences32_batched.json 2>/dev/null || true
python3 bench.py --sequences 64 --batched --group-size 16 --steps 40 > $OUT/${TAG}_bench_sequences64_batched.json 2>/dev/null || true
python3 bench.py --config kitti --sequences 21 --batched --steps 60 > $OUT/${TAG}_bench_kitti21_batched.json 2>/dev/null || true
python3 bench.py --config kitti --sequences 21 --batched --group-size 7 --steps 60 > $OUT/${TAG}_bench_kitti21_groups7.json 2>/dev/null || true
python3 bench.py --sequences 8 --runner-threads 2 --steps 40 > $OUT/${TAG}_bench_sequences8.json 2>/dev/null || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_dyn -- python3 $ROOT/bench.py --mode dynamic --steps 30 --no-cpu-baseline > $OUT/${TAG}_bench_dynamic_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_bat -- python3 $ROOT/bench.py $BATCHED > $OUT/${TAG}_bench_batched_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write_bat -- python3 $ROOT/bench.py $BATCHED > /dev/null 2>&1
cd $ROOT
# the --stats table of OUR kernels (the PyTorch kernels of the synthetic-image renderer filtered out)
for pair in "trace bench" "trace_dyn bench_dynamic" "trace_bat bench_batched"; do set -- $pair; f=$(ls $OUT/${TAG}_$1/*/*kernel_stats.csv | head -1); (head -1 $f; grep -v "at::\|Cijk\|elementwise\|rocclr\|^\"Name" $f) > $OUT/${TAG}_$2_kernel_stats.csv; done
python3 scripts/pmc_to_json.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.json
python3 scripts/pmc_to_json.py $OUT/${TAG}_pmc_fetch_bat $OUT/${TAG}_pmc_write_bat --batched 16 > $OUT/${TAG}_pmc_traffic_batched.json
python3 scripts/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_traffic.txt || true
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_trace_dyn $OUT/${TAG}_trace_bat $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_fetch_bat $OUT/${TAG}_pmc_write_bat
ls -la $OUT | grep ${TAG}
