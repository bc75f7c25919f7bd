"""Markdown tables of DESIGN.md §4 "current state" from a profile collection: usage: kernel_table.py <dir> <tag>  (e.g. profiles r04_b)
per workload: kernel | launches | avg µs | share of GPU time; and the PMC traffic of the default workload next to the algorithmic bytes where DESIGN states them."""
import csv, json, os, sys
d, tag = sys.argv[1], sys.argv[2]
ALG = {"be_solve_kernel": 0.87, "be_reduce_kernel": 2.4, "be_eval_kernel<true>": 2.3, "lk_track_kernel": 1.6, "gftt_tile_kernel": 1.0, "pyr_down_kernel": 4.9 / 3}      # MB per launch, DESIGN 4
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
def table(path, top):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["| kernel | launches | avg µs | share of our GPU time |", "|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| `%s` | %s | %.1f | %.1f %% |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    return "\n".join(out)
for name, title, top in (("bench", "default line (`bench.py --steps 30`, one sequence)", 18), ("bench_dynamic", "dynamic line (`--mode dynamic --steps 30`)", 22), ("bench_batched", "32 sequences in groups of 16 (`--sequences 32 --batched --group-size 16 --runner-threads 8 --steps 30`)", 16)):
    p = os.path.join(d, f"{tag}_{name}_kernel_stats.csv")
    if os.path.exists(p):
        print(f"\n**{title}** — `{p}`\n"); print(table(p, top))
for name, title in (("pmc_traffic", "default line"), ("pmc_traffic_batched", "32 sequences, 16 windows per launch")):
    p = os.path.join(d, f"{tag}_{name}.json")
    if not os.path.exists(p): continue
    j = json.load(open(p))
    print(f"\n**HBM-side traffic per launch, {title}** — `{p}` (FETCH_SIZE × 2 + WRITE_SIZE, separate `--pmc` passes, averages over active launches)\n")
    print("| kernel | fetch MB | write MB | traffic MB | algorithmic MB | ratio |"); print("|---|---|---|---|---|---|")
    w = j.get("windows_per_launch", 1)
    for k, v in sorted(j["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"]):
        base = k.replace("_batch_occ_kernel", "_kernel").replace("_batch_lm_kernel", "_kernel<true>").replace("_batch_kernel", "_kernel")      # (round 6: be_reduce_batch_occ_kernel, be_eval_batch_lm_kernel<8>)
        alg = next((a for n, a in ALG.items() if base.startswith(n)), None)
        alg = None if alg is None else alg * (w if "_batch" in k else 1)
        print("| `%s` | %.2f | %.2f | %.2f | %s | %s |" % (k, v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6, v["traffic_bytes"] / 1e6, "–" if alg is None else "%.1f" % alg, "–" if alg is None else "%.1f ×" % (v["traffic_bytes"] / 1e6 / alg)))
