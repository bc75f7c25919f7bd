cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gapprof; rocprofv3 --kernel-trace --output-format csv -d /tmp/gapprof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
f=$(ls /tmp/gapprof/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("lk_track_kernel")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("n", len(d), "first 12:", [round(x) for x in d[:12]])
s = sorted(d); print("p50 %.1f p90 %.1f p99 %.1f max %.1f" % (s[len(s)//2], s[int(len(s)*0.9)], s[int(len(s)*0.99)], s[-1]))
print("over 90:", [(i, round(x)) for i, x in enumerate(d) if x > 90][:20])
PY
