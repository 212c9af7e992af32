"""MFMA utilisation of the matrix-core panel kernels from a rocprofv3 --pmc pass (counters only, --kernel-trace):
   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d DIR -o p -- python scripts/config3_run.py
   python scripts/pmc_mfma.py DIR/p_results.db profiles/r06_pmc_mfma_configs2.md profiles/pmc_mfma.json configs2
Per kernel whose name contains "mfma": launches, average duration, the counters per launch (summed over the counter's
instances: one row per XCD / shader engine in the rocpd table) and
   busy_ratio = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES      (VERDICT r05 Next #3's figure: share of the SQ-busy time the matrix pipes were busy)
The json gets one entry per key (configs2, configs3) with the kernels' figures; bench.py attaches it to that config's roofline
object, labelled as coming from a separate profiled process."""
import json, os, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
out_md, out_json, key = sys.argv[2], sys.argv[3], sys.argv[4]
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
has_disp = "dispatch_id" in cols
# sum the instances of a counter within one dispatch, then average over dispatches
if has_disp:
    rows = db.execute("select name, counter_name, dispatch_id, sum(counter_value), max(duration) from pmc_events group by name, counter_name, dispatch_id").fetchall()
else:
    rows = [(n, c, i, v, d) for i, (n, c, v, d) in enumerate(db.execute("select name, counter_name, counter_value, duration from pmc_events"))]
by = {}
for name, c, disp, v, d in rows:
    k = re.sub(r"\(.*", "", name)
    e = by.setdefault(k, {})
    ce = e.setdefault(c, {"n": 0, "sum": 0.0, "dur": 0.0})
    ce["n"] += 1; ce["sum"] += v; ce["dur"] += d
lines = ["| kernel | launches | avg us | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | SQ_INSTS_VALU_MFMA_MOPS_F64 | MFMA busy / SQ busy |", "|---|---|---|---|---|---|---|"]
res = {}
for k, e in sorted(by.items(), key=lambda kv: -max(c["dur"] for c in kv[1].values())):
    g = lambda c: (e[c]["sum"] / e[c]["n"]) if c in e and e[c]["n"] else 0.0
    any_c = next(iter(e.values()))
    mf, bz, mops = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_INSTS_VALU_MFMA_MOPS_F64")
    ratio = mf / bz if bz else 0.0
    if "mfma" in k or mf > 0:
        res[k] = {"launches": any_c["n"], "avg_us": round(any_c["dur"] / any_c["n"] / 1e3, 2), "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CYCLES": bz,
                  "SQ_INSTS_VALU_MFMA_MOPS_F64": mops, "mfma_busy_over_sq_busy": round(ratio, 4)}
    lines.append(f"| `{k}` | {any_c['n']} | {any_c['dur'] / any_c['n'] / 1e3:.2f} | {mf:.4g} | {bz:.4g} | {mops:.4g} | {ratio:.4f} |")
open(out_md, "w").write("\n".join(lines[:40]) + "\n")
print("\n".join(lines[:16]))
allj = json.load(open(out_json)) if os.path.exists(out_json) else {}
allj[key] = {"kernels": res, "source": f"{os.path.basename(out_md)} (separate rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace pass over one solve)",
             "utilisation_is": "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per launch of the matrix-core panel kernel (counter instances summed); the panels are HBM-bound at every shape the solver uses (2 c flop per 8 B), so a low figure is expected: the matrix cores buy register footprint, not flop rate (DESIGN.md section 3)"}
json.dump(allj, open(out_json, "w"), indent=1)
