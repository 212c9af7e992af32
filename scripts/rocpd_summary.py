"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: calls, total/avg duration, share per kernel.
usage: python scripts/rocpd_summary.py <results.db> [out.md]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
span = db.execute("select min(start), max(end) from kernels").fetchone()
lines = ["| kernel | calls | total ms | avg us | min us | max us | % of kernel time |", "|---|---|---|---|---|---|---|"]
for name, calls, total, avg, mn, mx in rows:
    short = re.sub(r"\(.*", "", name)
    lines.append(f"| `{short}` | {calls} | {total/1e6:.2f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*total/tot:.1f} |")
lines.append("")
lines.append(f"kernel time total {tot/1e6:.1f} ms; first-to-last kernel span {(span[1]-span[0])/1e6:.1f} ms; "
             f"device busy {100*tot/(span[1]-span[0]):.1f} %")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
