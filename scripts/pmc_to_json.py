"""profiles/pmc_traffic.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only):
    python scripts/pmc_to_json.py <fetch_dir> <write_dir> > profiles/pmc_traffic.json
Averages over ACTIVE launches (duration > 8 us under the profiler: the trust-region schedule enqueues predicated no-op slots); FETCH_SIZE doubled
(gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section).  The file carries the digest of the kernel sources it was collected on;
bench.py reports roofline.traffic only while that digest matches."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest, git_head


def load(pat, name):
    d = collections.defaultdict(list)
    for p in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name:
                d[r["Kernel_Name"].split("(")[0].replace("void ", "", 1)].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d


BATCHED = len(sys.argv) > 3 and sys.argv[3] == "--batched"      # the passes ran `bench.py --sequences 16 --batched --group-size 8 --runner-threads 2` (argv[4] = windows per launch)
f = load(sys.argv[1] + "/**/*counter_collection.csv", "FETCH_SIZE")
w = load(sys.argv[2] + "/**/*counter_collection.csv", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --steps 30 --no-cpu-baseline; averages over active launches; "
                 "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); counters are kilobytes per dispatch",
       "git_head": git_head(), "csrc_digest": csrc_digest(), "kernels": {}}
if BATCHED:
    out["source"] = out["source"].replace("bench.py --steps 30 --no-cpu-baseline", "bench.py --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 30")
    out["windows_per_launch"] = int(sys.argv[4]) if len(sys.argv) > 4 else 8
for k in sorted(f):
    if BATCHED and "batch" not in k:
        continue
    if not (k.startswith(("be_", "lk_", "gftt_", "pyr_", "track_", "inst_", "roi_", "finalize", "compact", "lift", "erode")) or "be_" in k):
        continue
    fa = [v for v, t in f[k] if t > 8000] or [v for v, _ in f[k]]
    wa = [v for v, t in w.get(k, []) if t > 8000] or [v for v, _ in w.get(k, [(0, 0)])]
    fm, wm = 2 * sum(fa) / len(fa) * 1024.0, sum(wa) / len(wa) * 1024.0
    out["kernels"][k] = {"launches": len(f[k]), "active_launches": len(fa), "fetch_bytes_x2": round(fm, 1), "write_bytes": round(wm, 1), "traffic_bytes": round(fm + wm, 1)}
print(json.dumps(out, indent=1))
