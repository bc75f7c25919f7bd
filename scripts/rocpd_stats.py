"""Per-kernel summary (launches, total / average / min / max duration) of a `rocprofv3 --kernel-trace` run that wrote its default rocpd SQLite database
(`<dir>/<host>/<pid>_results.db`), as the CSV `--stats` prints for the csv output format.  usage: rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
dcols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
rows = cur.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for n, c, t, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n.replace('"', "'"), c, t, t / c, 100.0 * t / tot, mn, mx))
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
for ln in lines[:40]:
    print(ln[:200])
