"""Block JDQMR against the live reference (oracle/_ref) on the CPU checker: how far do the counts of eigs_jd.c sit from
dprimme's when the block size is > 1?  (Block size 1 reproduces them exactly: the control column.)  The reference's
block QMR indexes some recurrences by block position and others by original column (inner_solve.c:317, :373-377,
:616-620); eigs_jd.c keeps every recurrence per original column, so the runs are two different,
equally valid, block iterations.  Writes profiles/r03_jdqmr_block_count_sweep.txt.
    python scripts/jdqmr_block_sweep.py [nruns]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems
from checkers import eigsh, Operator

nruns = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
rows = []
for run in range(nruns):
    dims = [(30, 31), (24, 25, 3), (900,), (40, 41)][rng.integers(4)]
    rp, ci, va, n = problems.laplacian_csr(dims)
    b = int(rng.choice([1, 2, 4, 8]))
    method = str(rng.choice(["JDQMR", "JDQMR_ETol"]))
    target = str(rng.choice(["smallest", "largest", "closest_abs"]))
    kw = dict(numEvals=int(rng.integers(3, 9)), method=method, eps=float(rng.choice([1e-8, 1e-10])), maxBlockSize=b, target=target,
              iseed=tuple(int(x) for x in rng.integers(1, 4000, 4) | 1))
    if target == "closest_abs":
        kw["targetShifts"] = [float(rng.uniform(1.0, 3.0))]
    if rng.random() < 0.6:
        kw["precond"] = "jacobi" if target != "closest_abs" else ("fixed", kw["targetShifts"][0])
    out = {}
    for be in ("reference", "hostcheck"):
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend=be, dtype=np.float64, maxMatvecs=400000, **kw)
        out[be] = r
    a, h = out["reference"], out["hostcheck"]
    same = bool(np.max(np.abs(np.sort(a.evals) - np.sort(h.evals))) <= 1e-7 * a.params["aNorm"]) if a.ret == h.ret == 0 else False
    rows.append((b, method, target, "precond" in kw, n, kw["numEvals"], a.ret, h.ret, a.stats["numOuterIterations"], h.stats["numOuterIterations"],
                 a.stats["numMatvecs"], h.stats["numMatvecs"], same))
    print(rows[-1], flush=True)
lines = ["# Block JDQMR, product host solver over the CPU kernels vs the live reference: counts (scripts/jdqmr_block_sweep.py)", "",
         "| b | method | target | precond | n | numEvals | ret ref/ours | outer ref | outer ours | matvecs ref | matvecs ours | matvec ratio | same eigenvalues |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    lines.append(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]}/{r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[11] / max(1, r[10]):.3f} | {r[12]} |")
for b in (1, 2, 4, 8):
    sel = [r for r in rows if r[0] == b and r[6] == 0 and r[7] == 0]
    if sel:
        mv = np.array([r[11] / r[10] for r in sel]); ou = np.array([r[9] / r[8] for r in sel])
        lines.append("")
        lines.append(f"b = {b}: {len(sel)} runs, matvec ratio ours/ref min {mv.min():.3f} median {np.median(mv):.3f} max {mv.max():.3f}; "
                     f"outer-iteration ratio min {ou.min():.3f} median {np.median(ou):.3f} max {ou.max():.3f}")
open(os.path.join(ROOT, "profiles", "r03_jdqmr_block_count_sweep.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-8:]))
