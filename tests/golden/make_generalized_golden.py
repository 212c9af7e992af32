"""Generalised problems A x = lambda B x solved by the REAL reference (dprimme with massMatrixMatvec): tests/golden/
reference_generalized.json — eigenvalues, residual norms and counts for the parity tests of the mass-matrix path
(round 6).  A = 2-D / 3-D Laplacian, B = primme_amd.problems.mass_matrix_csr (closed form).
Run in the build container only (needs oracle/_ref):  python tests/golden/make_generalized_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems  # noqa: E402
from checkers import eigsh, Operator  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "gen_gdk": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="GD_plusK")),
    "gen_gdk_blk2": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="GD_plusK", maxBlockSize=2)),
    "gen_largest_soft": ((20, 21), dict(numEvals=4, eps=1e-9, aNorm=8.0, target="largest", locking=0)),
    "gen_olsen_jacobi": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="GD_Olsen_plusK", precond="jacobi")),
    "gen_gd": ((9, 10, 11), dict(numEvals=3, eps=1e-8, aNorm=12.0, method="GD")),
    "gen_lobpcg": ((9, 10, 11), dict(numEvals=3, eps=1e-7, aNorm=12.0, method="LOBPCG_OrthoBasis")),
    "gen_blk4_3d": ((12, 13, 14), dict(numEvals=8, eps=1e-9, aNorm=12.0, maxBlockSize=4)),
    "gen_noanorm": ((20, 21), dict(numEvals=3, eps=1e-8)),
    # the JDQMR inner solver with B (projectors on B Q and B x); locking on: without it the reference hands evecs over where
    # B evecs is meant (main_iter.c:334-336, :659-661) — "gen_jdqmr_soft" records that history too
    "gen_jdqmr": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="JDQMR", locking=1)),
    "gen_jdqmr_jacobi": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="JDQMR", precond="jacobi", locking=1)),
    "gen_jdqmr_etol_3d": ((9, 10, 11), dict(numEvals=4, eps=1e-9, aNorm=12.0, method="JDQMR_ETol", precond="jacobi", locking=1)),
    "gen_jdqmr_largest": ((20, 21), dict(numEvals=4, eps=1e-9, aNorm=8.0, method="JDQMR", target="largest", locking=1)),
    "gen_jdqmr_blk3": ((12, 13, 14), dict(numEvals=6, eps=1e-9, aNorm=12.0, method="JDQMR_ETol", maxBlockSize=3, precond="jacobi", locking=1)),
    "gen_jd_olsen": ((20, 21), dict(numEvals=4, eps=1e-9, aNorm=8.0, method="JD_Olsen_plusK", precond="jacobi")),
    "gen_jdqmr_soft": ((20, 21), dict(numEvals=5, eps=1e-9, aNorm=8.0, method="JDQMR", precond="jacobi", locking=0)),
    # the classic pair of the reference's own data files: LUNDA.mtx x = lambda lund_b.mtx x (n = 147, both s.p.d.; reference_driver/)
    "gen_lund_gdk": ("lund", dict(numEvals=5, eps=1e-10, method="GD_Olsen_plusK", precond="jacobi")),
    "gen_lund_jdqmr": ("lund", dict(numEvals=5, eps=1e-10, method="JDQMR", precond="jacobi", locking=1)),
    "gen_lund_blk2": ("lund", dict(numEvals=4, eps=1e-10, maxBlockSize=2, precond="jacobi", locking=1)),
}


def operators(dims):
    if dims == "lund":
        data = os.path.join(HERE, "reference_driver")
        rp, ci, va, n, _ = problems.read_matrix_market(os.path.join(data, "LUNDA.mtx"))
        brp, bci, bva, nb, _ = problems.read_matrix_market(os.path.join(data, "lund_b.mtx"))
        assert n == nb
        return (rp, ci, va), (brp, bci, bva), n
    rp, ci, va, n = problems.laplacian_csr(tuple(dims))
    return (rp, ci, va), problems.mass_matrix_csr(n), n


def main():
    out = {}
    for name, (dims, kw) in CASES.items():
        (rp, ci, va), (brp, bci, bva), n = operators(dims)
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", mass=Operator(n, csr=(brp, bci, bva)), v0=problems.start_vector(n), **kw)
        out[name] = dict(dims=dims if isinstance(dims, str) else list(dims), kwargs=kw, ret=r.ret, initSize=r.initSize, evals=r.evals.tolist(), resNorms=r.resNorms.tolist(),
                         params=r.params, stats={k: r.stats[k] for k in ("numOuterIterations", "numMatvecs", "numRestarts", "numPreconds")})
        print(name, r.ret, out[name]["stats"])
    json.dump(out, open(os.path.join(HERE, "reference_generalized.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
