"""Where the device idles: for every (previous kernel -> next kernel) pair of a rocprofv3 (rocpd sqlite)
kernel trace, the number of occurrences and the mean idle time between the end of one and the start of
the other.  Only the steady part of the trace (gaps below --max-gap-us) is counted.
usage: python scripts/gap_analysis.py <results.db> [out.md] [--max-gap-us 2000]"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
maxgap = 2000.0
if "--max-gap-us" in sys.argv:
    maxgap = float(sys.argv[sys.argv.index("--max-gap-us") + 1])
db = sqlite3.connect(args[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
pairs = {}
tot_gap = 0.0
tot_busy = 0.0
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    gap = (s1 - e0) / 1e3
    tot_busy += (e0 - s0) / 1e3
    if gap > maxgap:
        continue
    g = max(gap, 0.0)
    tot_gap += g
    key = (short(n0), short(n1))
    c = pairs.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += g
lines = ["| previous kernel | next kernel | count | mean gap us | total gap ms | % of all idle |", "|---|---|---|---|---|---|"]
for (a, b), (cnt, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    lines.append(f"| `{a}` | `{b}` | {cnt} | {g/cnt:.2f} | {g/1e3:.2f} | {100*g/tot_gap:.1f} |")
lines.append("")
lines.append(f"kernels {len(rows)}; busy {tot_busy/1e3:.1f} ms; idle between kernels (gaps < {maxgap:.0f} us) {tot_gap/1e3:.1f} ms "
             f"= {100*tot_gap/(tot_gap+tot_busy):.1f} % of busy + idle")
out = "\n".join(lines)
print(out)
if len(args) > 1:
    open(args[1], "w").write(out + "\n")
