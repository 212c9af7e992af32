"""BASELINE configs[2]: LUNDA.mtx tiled block-diagonally T times, tile t scaled by
1 + t/T, 20 eigenvalues closest_abs to a shift (default 4.4764e8), JDQMR, block size 8, eps 1e-8 |A|, Jacobi
K = diag(A) - shift.  Truth: union of the scaled dense spectra of the 147 x 147 tile.

    python scripts/config3_run.py [--tiles 34014] [--backend hip|hostcheck|reference] [--prof]
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=34014)
    ap.add_argument("--backend", default="hip")
    ap.add_argument("--method", default="JDQMR")
    ap.add_argument("--block", type=int, default=8)
    ap.add_argument("--num-evals", type=int, default=20)
    ap.add_argument("--shift", type=float, default=4.4764e8)   # see DESIGN.md §6 for why not SURVEY's 1.0e6
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--precond", default="fixed", choices=["fixed", "davidson", "none"])
    ap.add_argument("--scale-range", type=float, default=1.0)
    ap.add_argument("--max-matvecs", type=int, default=0)
    ap.add_argument("--reps", type=int, default=1, help="solves; the fastest is reported")
    args = ap.parse_args()
    from primme_amd import problems, _ffi as F
    from checkers import Operator, Session
    import reference_driver_cases as RD
    rp, ci, va, n0 = RD.lunda()
    T = args.tiles
    t0 = time.time()
    trp, tci, tva = problems.tile_block_diagonal(rp, ci, va, T, lambda t: 1.0 + args.scale_range * t / T)
    A = np.zeros((n0, n0)); A[np.repeat(np.arange(n0), np.diff(rp)), ci] = va
    w0 = np.linalg.eigvalsh(A)
    w = (w0[None, :] * (1.0 + args.scale_range * np.arange(T) / T)[:, None]).ravel()
    aNorm = float(np.abs(w).max())
    want = np.sort(w[np.argsort(np.abs(w - args.shift))][:args.num_evals])
    gaps = np.diff(np.sort(w[np.argsort(np.abs(w - args.shift))][:args.num_evals + 5]))
    n = n0 * T
    print(f"n={n} nnz={len(tva)} build {time.time()-t0:.1f}s aNorm={aNorm:.4e} min gap near shift={gaps.min():.3e} tol={args.eps*aNorm:.3e}", flush=True)
    op = Operator(n, csr=(trp, tci, tva))
    sess = Session(op, backend=args.backend)
    kw = dict(numEvals=args.num_evals, target="closest_abs", targetShifts=[args.shift], method=args.method,
              maxBlockSize=args.block, eps=args.eps, aNorm=aNorm, precond={"fixed": ("jacobi", args.shift), "davidson": "jacobi", "none": None}[args.precond], return_evecs=False)
    if args.max_matvecs: kw["maxMatvecs"] = args.max_matvecs
    if args.prof and args.backend == "hip":
        sess.lib.hipk_prof_reset(); sess.lib.hipk_prof_enable(1)
    el = None
    for _ in range(max(1, args.reps)):
        t0 = time.time()
        r = sess.solve(**kw)
        el = min(el, time.time() - t0) if el is not None else time.time() - t0
    out = dict(n=n, tiles=T, ret=r.ret, seconds=round(el, 3), eigenpairs_per_s=round(args.num_evals / el, 4),
               outer=r.stats["numOuterIterations"], matvecs=r.stats["numMatvecs"], restarts=r.stats["numRestarts"],
               preconds=r.stats["numPreconds"], maxBasisSize=r.params["maxBasisSize"],
               max_eval_err=float(np.max(np.abs(np.sort(r.evals) - want))) if r.ret == 0 else None,
               max_resnorm=float(r.resNorms.max()), tol=args.eps * aNorm,
               time_matvec=r.stats["timeMatvec"], time_precond=r.stats["timePrecond"], time_ortho=r.stats["timeOrtho"])
    if args.prof and args.backend == "hip":
        sess.lib.hipk_prof_enable(0)
        names = ["dots", "project", "ritz", "spmv"]
        for c in range(4):
            ms, ln, nb = C.c_double(), C.c_long(), C.c_double()
            sess.lib.hipk_prof_get(c, C.byref(ms), C.byref(ln), C.byref(nb))
            out["prof_" + names[c]] = dict(ms=round(ms.value, 2), launches=ln.value, GBps=round(nb.value / max(ms.value, 1e-9) / 1e6, 1))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
