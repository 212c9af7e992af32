"""Where the device sits idle in a rocprofv3 (rocpd sqlite) kernel trace: the gap before each kernel launch
(start - end of the previous kernel on the device), summed per (previous kernel -> next kernel) pair.
Gaps longer than --cut ms (set-up, between solves) are left out.
usage: python scripts/rocpd_gaps.py <results.db> [out.md] [--cut 20]"""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cut = 20.0
if "--cut" in sys.argv: cut = float(sys.argv[sys.argv.index("--cut") + 1]); args = [a for a in args if a != str(cut) and a != sys.argv[sys.argv.index("--cut") + 1]]
db = sqlite3.connect(args[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"^void ", "", re.sub(r"\(.*", "", n))[:60]
pair, nxt = {}, {}
busy = idle = 0
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    g = max(0, s1 - e0)
    if g > cut * 1e6: continue
    busy += e1 - s1; idle += g
    k = (short(n0), short(n1))
    c = pair.setdefault(k, [0, 0]); c[0] += 1; c[1] += g
    c = nxt.setdefault(short(n1), [0, 0]); c[0] += 1; c[1] += g
lines = [f"kernel busy {busy/1e6:.1f} ms, idle between kernels {idle/1e6:.1f} ms (gaps > {cut} ms left out)", "",
         "| idle before kernel | launches | total idle ms | avg us |", "|---|---|---|---|"]
for k, (c, g) in sorted(nxt.items(), key=lambda kv: -kv[1][1])[:20]:
    lines.append(f"| `{k}` | {c} | {g/1e6:.2f} | {g/1e3/c:.1f} |")
lines += ["", "| previous -> next | count | total idle ms | avg us |", "|---|---|---|---|"]
for k, (c, g) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:30]:
    lines.append(f"| `{k[0]}` -> `{k[1]}` | {c} | {g/1e6:.2f} | {g/1e3/c:.1f} |")
out = "\n".join(lines)
print(out)
if len(args) > 1: open(args[1], "w").write(out + "\n")
