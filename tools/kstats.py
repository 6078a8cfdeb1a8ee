"""python tools/kstats.py <rocprofv3 results .db> [csv out]: per-kernel call count, total / mean / min / max duration from the
rocpd SQLite file `rocprofv3 --kernel-trace` writes (the --stats summary of ROCm 7 lives in the same file)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
dispatch = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
symbols = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in cur.execute(f"pragma table_info({symbols})")]
name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
rows = cur.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                   f"from {dispatch} d join {symbols} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for name, calls, tot, lo, hi in rows:
    short = name.split("(")[0]
    lines.append(f"\"{short}\",{calls},{tot},{tot / calls:.1f},{100.0 * tot / total:.2f},{lo},{hi}")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
