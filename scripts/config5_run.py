"""BASELINE configs[4] on ONE GPU: A 8 000 000 x 2 000 000, row i has 5 nonzeros at
columns (i*p_q + q) mod n, values 1 + ((i+q) mod 13)/13; 10 largest singular triplets through the
normal equations (hip_dprimme_svds, GD+k on A'A).

    python scripts/config5_run.py [--rows 8000000] [--cols 2000000] [--backend hip|hostcheck|reference]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8_000_000)
    ap.add_argument("--cols", type=int, default=2_000_000)
    ap.add_argument("--num-svals", type=int, default=10)
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--backend", default="hip")
    ap.add_argument("--method", default="GD_plusK")
    ap.add_argument("--max-matvecs", type=int, default=0)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    from primme_amd import problems
    from checkers import svds, transpose_csr
    t0 = time.time()
    rp, ci, va = problems.svds_synthetic_csr(args.rows, args.cols)
    print(f"m={args.rows} n={args.cols} nnz={len(va)} build {time.time()-t0:.1f}s", flush=True)
    t0 = time.time()
    r = svds(args.rows, args.cols, (rp, ci, va), numSvals=args.num_svals, eps=args.eps, methodStage1=args.method,
             backend=args.backend, maxMatvecs=args.max_matvecs, return_vectors=args.check)
    el = time.time() - t0
    out = dict(m=args.rows, n=args.cols, ret=r.ret, seconds_incl_upload=round(el, 3), solver_seconds=round(r.stats["elapsedTime"], 3),
               triplets_per_s=round(args.num_svals / max(r.stats["elapsedTime"], 1e-9), 4),
               outer=r.stats["numOuterIterations"], matvecs=r.stats["numMatvecs"], restarts=r.stats["numRestarts"],
               svals=[float(x) for x in r.svals], max_resnorm=float(r.resNorms.max()) if len(r.resNorms) else None,
               tol=args.eps * r.params["aNorm"], aNorm=r.params["aNorm"])
    if args.check and r.U is not None:
        rpT, ciT, vaT = transpose_csr(args.rows, args.cols, rp, ci, va)
        AV = problems.csr_matvec_numpy(rp, ci, va, r.V)
        AtU = problems.csr_matvec_numpy(rpT, ciT, vaT, r.U)
        out["true_residual_max"] = float(np.sqrt(np.sum((AV - r.U * r.svals) ** 2, axis=0) + np.sum((AtU - r.V * r.svals) ** 2, axis=0)).max())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
