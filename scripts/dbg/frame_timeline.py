"""debug: timeline of one steady-state frame from a rocprofv3 --kernel-trace CSV (iteration kernels folded into one line per iteration)"""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(p))]
ours = [r for r in rows if any(t in r["Kernel_Name"] for t in ("be_", "lk_", "gftt", "pyr_", "track_", "remap"))]
ours.sort(key=lambda r: int(r["Start_Timestamp"]))
g = [i for i, r in enumerate(ours) if "be_gauge" in r["Kernel_Name"]]
i0, i1 = g[len(g) // 2], g[len(g) // 2 + 1]
t0 = int(ours[i0]["Start_Timestamp"])
it = 0
for r in ours[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    if nm.startswith(("be_reduce", "be_solve")) or "be_eval_kernel<true>" in r["Kernel_Name"]:
        if nm.startswith("be_eval"): it += 1
        if it not in (1, 10): continue
    print("%-30s q%s start %8.1f dur %6.1f" % (nm, r["Queue_Id"], (s - t0) / 1e3, (e - s) / 1e3))
print("frame period %.1f us" % ((int(ours[i1]["Start_Timestamp"]) - t0) / 1e3))
