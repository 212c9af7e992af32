"""Random configurations over methods, targets, block sizes, preconditioner, extraction: the product host solver (hostcheck)
against the live reference -- return codes, eigenvalues, residual level; counts are reported, not required to match
(block methods and interior targets are rounding-sensitive in both codes).  CPU only.
usage: python scripts/parity_sweep_broad.py <seed> <cases>
MASS=1: every case is a generalised problem A x = lambda B x (B = problems.mass_matrix_csr, randomly scaled); JDQR is left out (the live
reference fails there) and the counts compared are outer iterations and restarts (B is applied on demand here)."""
import sys, os, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import checkers
from checkers import eigsh
from primme_amd import problems, Operator
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
METHODS = ["DEFAULT_MIN_TIME", "DEFAULT_MIN_MATVECS", "GD", "GD_plusK", "GD_Olsen_plusK", "JD_Olsen_plusK", "JDQR", "JDQMR", "JDQMR_ETol", "LOBPCG_OrthoBasis", "LOBPCG_OrthoBasis_Window"]
bad = cnt_diff = ran = 0
MASS = bool(os.environ.get("MASS"))
if MASS: METHODS = [m for m in METHODS if m != "JDQR"]
for t in range(N):
    dims = tuple(int(x) for x in rng.integers(6, 18, size=rng.integers(1, 4)))
    rp, ci, va, n = problems.laplacian_csr(dims)
    if n < 80: continue
    nev = int(rng.integers(1, 9))
    method = str(rng.choice(METHODS))
    kw = dict(numEvals=nev, eps=float(10.0 ** -rng.integers(5, 10)), aNorm=4.0 * len(dims), v0=problems.start_vector(n), method=method,
              target=str(rng.choice(["smallest", "largest", "closest_abs", "closest_geq"])))
    if kw["target"] in ("closest_abs", "closest_geq"):
        kw["targetShifts"] = [float(rng.uniform(0.5, 3.5 * len(dims)))]
    if rng.random() < 0.5: kw["maxBlockSize"] = int(rng.integers(1, 5))
    if rng.random() < 0.4: kw["precond"] = "jacobi"
    if rng.random() < 0.3: kw["locking"] = int(rng.integers(0, 2))
    if rng.random() < 0.25: kw["projection"] = str(rng.choice(["refined", "harmonic"])) if kw["target"] in ("closest_abs", "closest_geq") else "RR"
    kw["maxMatvecs"] = 20000
    if MASS:
        brp, bci, bva = problems.mass_matrix_csr(n)
        kw["mass"] = Operator(n, csr=(brp, bci, bva * float(rng.choice([1.0, 0.25, 7.0]))))
        kw["projection"] = "RR"
        if rng.random() < 0.5: kw.pop("aNorm")          # the estimates of |A|, |B|, |B^-1| then decide the stopping rule
    try:
        h = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", **kw)
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", **kw)
    except Exception as e:
        print("EXC", dims, {k: v for k, v in kw.items() if k != "v0"}, repr(e)[:200]); bad += 1; continue
    ran += 1
    desc = (dims, {k: v for k, v in kw.items() if k not in ("v0", "mass")})
    if h.ret != r.ret:
        if {h.ret, r.ret} <= {0, -3}:
            print("RET", desc, h.ret, r.ret, h.stats["numMatvecs"], r.stats["numMatvecs"])
        else:
            bad += 1; print("BADRET", desc, h.ret, r.ret)
        continue
    if h.ret == 0:
        tol = kw["eps"] * (kw["aNorm"] if "aNorm" in kw else r.params["aNorm"] * max(1.0, r.stats.get("estimateInvBNorm", 1.0)))
        if len(h.evals) != len(r.evals) or np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) > 10 * tol:
            bad += 1; print("EVALS", desc, np.sort(h.evals), np.sort(r.evals))
        if np.any(h.resNorms > 2 * max(tol, r.resNorms.max())):
            bad += 1; print("RESN", desc, h.resNorms.max(), tol)
    keys = ("numOuterIterations", "numRestarts") if MASS else ("numOuterIterations", "numMatvecs", "numRestarts")
    if any(h.stats[k] != r.stats[k] for k in keys):
        cnt_diff += 1
        big = "numOuterIterations" if MASS else "numMatvecs"
        rel = abs(h.stats[big] - r.stats[big]) / max(1, r.stats[big])
        if rel > 0.2: print("COUNT", desc, [h.stats[k] for k in keys], [r.stats[k] for k in keys])
print("ran", ran, "bad", bad, "count differences", cnt_diff)
