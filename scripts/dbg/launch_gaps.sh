#!/bin/bash
# debug: gaps between consecutive kernels of the BA chain (end of one -> start of the next) from a rocprofv3 kernel trace of the default bench
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gapprof; rocprofv3 --kernel-trace --output-format csv -d /tmp/gapprof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-dynamic-line > /dev/null 2>&1
f=$(ls /tmp/gapprof/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows = [r for r in rows if not r["Kernel_Name"].startswith(("at::", "void at::", "Cijk", "void (anonymous")) and "elementwise" not in r["Kernel_Name"]]
byq = collections.defaultdict(list)
for r in rows: byq[r.get("Queue_Id", "0")].append(r)
pairs = []
for q, rs in byq.items():
    rs.sort(key=lambda r: int(r["Start_Timestamp"]))
    pairs += list(zip(rs, rs[1:]))
gaps = collections.defaultdict(list)
for a, b in pairs:
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    na, nb = a["Kernel_Name"].split("(")[0].replace("void ", ""), b["Kernel_Name"].split("(")[0].replace("void ", "")
    if g < 50000: gaps[(na[:28], nb[:28])].append(g)
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 20: print(f"{k[0]:30s} -> {k[1]:30s} n {len(v):5d}  mean {sum(v)/len(v)/1e3:6.2f} us  min {min(v)/1e3:6.2f}")
PY
python3 - "$f" <<'PY'
import csv, sys
seen = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
    if n.startswith(("be_", "lk_", "pyr_", "gftt", "track_")) and n not in seen:
        seen[n] = (r["Scratch_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"])
for n, v in seen.items(): print(f"{n:36s} scratch {v[0]:>6s} lds {v[1]:>7s} vgpr {v[2]:>4s} agpr {v[3]:>4s} sgpr {v[4]:>4s} wg {v[5]:>5s} grid {v[6]}")
PY
