"""per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kilobytes per dispatch).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at 64 B -> doubled."""
import csv, sys, collections, glob
def load(pat, name):
    d = collections.defaultdict(list)
    for p in glob.glob(pat):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name:
                d[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d
f = load(sys.argv[1] + "/*/*counter_collection.csv", "FETCH_SIZE")
w = load(sys.argv[2] + "/*/*counter_collection.csv", "WRITE_SIZE")
print(f"{'kernel':44s} {'n':>6s} {'n_act':>6s} {'fetch_KB(x2)':>13s} {'write_KB':>9s} {'traffic_KB':>10s}   (averages over ACTIVE launches: duration > 8 us under the profiler)")
for k in sorted(f, key=lambda k: -sum(v for v, _ in f[k])):
    if not (k.startswith("be_") or "be_" in k or k.startswith(("lk_", "gftt_", "pyr_", "track_"))):
        continue
    fa = [v for v, t in f[k] if t > 8000] or [v for v, _ in f[k]]
    wa = [v for v, t in w.get(k, []) if t > 8000] or [v for v, _ in w.get(k, [(0, 0)])]
    fm, wm = 2 * sum(fa) / len(fa), sum(wa) / len(wa)
    print(f"{k[:44]:44s} {len(f[k]):6d} {len(fa):6d} {fm:13.1f} {wm:9.1f} {fm + wm:10.1f}")
