#!/bin/bash
# on the GPU box: FETCH_SIZE / WRITE_SIZE of the default workload's kernels (two passes) -> per-kernel traffic table on stdout
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p1f /tmp/p1w
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p1f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p1w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line --no-extra-lines > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 scripts/pmc_summary.py /tmp/p1f /tmp/p1w
