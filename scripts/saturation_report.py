"""What the GPU runs out of with S sequences per GPU: one table from three rocprofv3 passes of `bench.py --sequences S --batched` (scripts/saturation_probe.sh).
    python scripts/saturation_report.py <trace_dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_l2_dir>] <frames_per_s_unprofiled> <S>
  * kernel trace (timestamps): over the steady-state part of the run — the union of busy time, the mean number of OUR kernels in flight, per kernel name the launches,
    average duration and share of the summed kernel time; per hardware queue the gaps between one kernel's end and the next one's start
  * FETCH_SIZE / WRITE_SIZE passes (kernels serialised by the profiler: per-launch bytes are what the kernel moves on its own): bytes per frame -> HBM GB/s at the
    UNPROFILED frame rate, against 8 TB/s (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md)
  * TCC_HIT / TCC_MISS pass (optional): L2 hit rate per kernel
Prints JSON."""
import collections, csv, glob, json, sys


def ours(name):
    n = name.replace("void ", "", 1)
    return n.startswith(("be_", "lk_", "gftt_", "pyr_", "track_", "inst_", "roi_", "xp_", "bd::", "undist", "erode", "circle", "viode")) or "be_" in n


def short(name):
    return name.split("(")[0].replace("void ", "", 1)


def trace(d):
    rows = []
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if ours(r["Kernel_Name"]):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    if not rows:
        return {}
    t_lo = rows[len(rows) // 3][0]                       # skip the warm-up third
    rows = [r for r in rows if r[0] >= t_lo]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _, _ in rows)
    per = collections.defaultdict(list)
    for s, e, n, _ in rows:
        per[n].append(e - s)
    gaps = collections.defaultdict(list)
    last = {}
    for s, e, n, q in rows:
        if q in last and s >= last[q]:
            gaps[q].append(s - last[q])
        last[q] = max(last.get(q, 0), e)
    allg = sorted(g for v in gaps.values() for g in v)
    pct = lambda v, p: v[min(len(v) - 1, int(p * len(v)))] if v else None
    return {"window_ms": span / 1e6, "fraction_of_time_with_a_kernel_of_ours_running": busy / span, "mean_kernels_in_flight": tot / span, "hardware_queues_seen": len(gaps),
            "gap_between_kernels_on_one_queue_us": {"p50": pct(allg, 0.5) / 1e3 if allg else None, "p90": pct(allg, 0.9) / 1e3 if allg else None, "n": len(allg)},
            "kernels": {n: {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "share_of_kernel_time": sum(v) / tot} for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]}}


def counters(d, name):
    out = collections.defaultdict(list)
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name and ours(r["Kernel_Name"]):
                out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


def main():
    a = sys.argv[1:]
    tr, fdir, wdir = a[0], a[1], a[2]
    l2dir = a[3] if len(a) == 6 else None
    fps, S = float(a[-2]), int(a[-1])
    rep = {"sequences_per_gpu": S, "frames_per_s_unprofiled": fps, "trace": trace(tr)}
    f, w = counters(fdir, "FETCH_SIZE"), counters(wdir, "WRITE_SIZE")
    fb = sum(2 * sum(v) * 1024.0 for v in f.values()); wb = sum(sum(v) * 1024.0 for v in w.values())
    launches = sum(len(v) for v in f.values())
    # frames covered by the profiled run: warm-up 12 + 2 x steps per sequence (the probe passes --steps 30)
    frames = S * (12 + 60)
    rep["hbm"] = {"bytes_per_frame": (fb + wb) / frames, "GBs_at_the_unprofiled_rate": (fb + wb) / frames * fps / 1e9, "fraction_of_8TBs": (fb + wb) / frames * fps / 8e12,
                  "kernel_launches_profiled": launches, "note": "FETCH_SIZE x 2 + WRITE_SIZE over every launch of our kernels in the profiled run / frames of the run"}
    if l2dir:
        h, m = counters(l2dir, "TCC_HIT_sum"), counters(l2dir, "TCC_MISS_sum")
        tot_h, tot_m = sum(sum(v) for v in h.values()), sum(sum(v) for v in m.values())
        rep["l2"] = {"hit_rate_all_kernels": tot_h / max(tot_h + tot_m, 1.0),
                     "per_kernel": {k: round(sum(h[k]) / max(sum(h[k]) + sum(m.get(k, [0])), 1.0), 3) for k in sorted(h, key=lambda k: -sum(h[k]) - sum(m.get(k, [0])))[:10]}}
    print(json.dumps(rep, indent=1))


main()
