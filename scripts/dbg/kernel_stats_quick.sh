cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lkprof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lkprof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
f=$(ls /tmp/lkprof/*/*kernel_stats.csv | head -1); grep -E "lk_track|gftt|pyr_|be_" $f | sed "s/([^)]*)//" | cut -c1-110
