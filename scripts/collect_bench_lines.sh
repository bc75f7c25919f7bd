#!/bin/bash
# the bench lines of scripts/collect_profiles.sh alone (no profiler passes): usage scripts/collect_bench_lines.sh <tag> -> gpurun_out/<tag>_bench*.json
TAG=${1:-r04_c}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
run() { name=$1; shift; python3 bench.py "$@" > $OUT/${TAG}_bench${name}.json 2> $OUT/${TAG}_bench${name}.err || echo "bench${name}: exit $? (no line: see ${TAG}_bench${name}.err)"; }
run ""                      
run _steps20                --steps 20 --warmup 5
run _dynamic                --mode dynamic --no-cpu-baseline
run _sequences16_batched    --sequences 16 --batched --steps 60
run _sequences16_groups8    --sequences 16 --batched --group-size 8 --steps 60
run _sequences16_teams      --sequences 16 --batched --teams --runner-threads 4 --steps 60
run _sequences32_groups8    --sequences 32 --batched --group-size 8 --steps 40
run _sequences32_batched    --sequences 32 --batched --group-size 16 --steps 40
run _sequences64_batched    --sequences 64 --batched --group-size 16 --steps 40
run _kitti21_batched        --config kitti --sequences 21 --batched --steps 60
run _kitti21_groups7        --config kitti --sequences 21 --batched --group-size 7 --steps 60
run _sequences8             --sequences 8 --runner-threads 2 --steps 40
for f in $OUT/${TAG}_bench*.err; do [ "$(grep -v amdgpu.ids $f | wc -c)" -le 1 ] && rm -f $f; done
ls $OUT | grep ${TAG}_
