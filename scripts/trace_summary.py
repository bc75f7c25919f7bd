"""summarise a rocprofv3 --kernel-trace CSV: per kernel count / total / avg, and the same over 'active' launches only
(the trust-region schedule enqueues a fixed number of slots; launches after convergence return immediately)."""
import csv, sys, collections, glob
paths = sys.argv[1:] or glob.glob("gpurun_out/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for p in paths:
    for r in csv.DictReader(open(p)):
        nm = r["Kernel_Name"]
        if "at::" in nm or "elementwise" in nm or "Cijk" in nm or "reduce_kernel<" in nm:
            continue
        d[nm.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in d.items():
    act = [x for x in v if x > 4.0]
    rows.append((sum(v), k, len(v), sum(v) / len(v), len(act), (sum(act) / len(act)) if act else 0.0, max(v)))
print(f"{'kernel':60s} {'n':>7s} {'total_ms':>9s} {'avg_us':>8s} {'n_act':>7s} {'act_us':>8s} {'max_us':>8s}")
for t, k, n, avg, na, aavg, mx in sorted(rows, reverse=True):
    print(f"{k:60s} {n:7d} {t/1e3:9.2f} {avg:8.1f} {na:7d} {aavg:8.1f} {mx:8.1f}")
