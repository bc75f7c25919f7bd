"""timeline of the multi-sequence regime from a rocprofv3 --kernel-trace CSV: per-kernel averages, and ONE steady-state round printed launch by launch
(start offset, duration, queue) — which launches overlap, where the gaps are.  usage: round_timeline.py <dir> [anchor kernel substring] [round index]"""
import csv, glob, sys, collections
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "pyr_down_multi"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Kernel_Name"]
    if "at::" in nm or "elementwise" in nm or "Cijk" in nm or "reduce_kernel<" in nm or "rocclr" in nm:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:44], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
agg = collections.defaultdict(list)
for s, e, n, q, g, wg in rows:
    agg[n].append((e - s) / 1e3)
print(f"{'kernel':46s} {'n':>6s} {'avg_us':>8s} {'p50':>7s} {'max':>8s} {'total_ms':>9s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{n:46s} {len(v):6d} {sum(v)/len(v):8.1f} {v2[len(v2)//2]:7.1f} {v2[-1]:8.1f} {sum(v)/1e3:9.2f}")
# one round: from the `which`-th launch of the anchor kernel (first of its burst) to the next burst
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
starts = [i for k, i in enumerate(idx) if k == 0 or rows[i][0] - rows[idx[k - 1]][0] > 300000]      # bursts separated by > 0.3 ms
if len(starts) > abs(which) + 1:
    a, b = starts[which], starts[which + 1]
    t0 = rows[a][0]
    print(f"\none round ({(rows[b][0] - t0) / 1e3:.1f} us between two bursts of {anchor}): start_us dur_us queue grid wg kernel")
    busy_end = t0
    for s, e, n, q, g, wg in rows[a:b]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:>3s} {g:>8s} {wg:>5s}  {n}")
