"""Per-kernel averages of the PMC counters in a rocprofv3 (rocpd sqlite) result:
usage: python scripts/pmc_summary.py <results.db> [out.md]
Columns: launches, average duration, and for every collected counter its per-launch average."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                  "group by name, counter_name").fetchall()
ctrs = sorted({r[1] for r in rows})
by = {}
for name, c, n, v, d in rows:
    e = by.setdefault(re.sub(r"\(.*", "", name), {"n": n, "dur": d})
    e[c] = v
lines = ["| kernel | launches | avg us | " + " | ".join(ctrs) + " |", "|---|---|---|" + "---|" * len(ctrs)]
for k, e in sorted(by.items(), key=lambda kv: -kv[1]["n"] * kv[1]["dur"]):
    lines.append(f"| `{k}` | {e['n']} | {e['dur']/1e3:.2f} | " + " | ".join(f"{e.get(c, 0):.4g}" for c in ctrs) + " |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
