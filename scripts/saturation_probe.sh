#!/bin/bash
# On the GPU box: what S sequences per GPU cost the device (scripts/saturation_report.py).  usage: scripts/saturation_probe.sh <tag> <S>  -> gpurun_out/<tag>_saturation_S<S>.json
set -e
TAG=${1:-r05}; S=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=12 BENCH_BIT_IDENTITY=0
CMD="--sequences $S --batched --steps 30"
cd $ROOT
FPS=$(python3 bench.py $CMD 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sat_*
rocprofv3 --kernel-trace --output-format csv -d /tmp/sat_trace -- python3 $ROOT/bench.py $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sat_fetch -- python3 $ROOT/bench.py $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sat_write -- python3 $ROOT/bench.py $CMD > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/sat_l2 -- python3 $ROOT/bench.py $CMD > /dev/null 2>&1 || rm -rf /tmp/sat_l2
if [ -d /tmp/sat_l2 ]; then L2=/tmp/sat_l2; else L2=; fi
python3 $ROOT/scripts/saturation_report.py /tmp/sat_trace /tmp/sat_fetch /tmp/sat_write $L2 $FPS $S > $OUT/${TAG}_saturation_S${S}.json
cat $OUT/${TAG}_saturation_S${S}.json | head -60
