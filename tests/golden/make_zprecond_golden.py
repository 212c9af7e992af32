"""Generates tests/golden/reference_zprecond.json: the REAL reference's zprimme (oracle/_ref/libprimme_ref.so) on the problem
of examples/ex_eigs_zhip_precond.hip — Hermitian tridiagonal matrix, JDQMR with a NON-Hermitian complex diagonal
preconditioner, so that x'K^-1 x of the skew projector (reference src/eigs/correction.c:969-977, inner_solve.c:737-741) is
complex.  Run in the build container only:  python tests/golden/make_zprecond_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems  # noqa: E402
from checkers import eigsh, Operator  # noqa: E402

N, GAMMA = 2000, 0.1
CASES = {"jdqmr": dict(method="JDQMR"), "jdqmr_etol": dict(method="JDQMR_ETol"), "gd_olsen": dict(method="GD_Olsen_plusK")}


def main():
    rp, ci, va, d, a = problems.hermitian_tridiag_graded(N)
    out = {"n": N, "gamma": GAMMA, "cases": {}}
    for name, kw in CASES.items():
        r = eigsh(Operator(N, csr=(rp, ci, va)), backend="reference", dtype=np.complex128, numEvals=4, eps=1e-10, aNorm=2001.0, maxMatvecs=20000,
                  precond=("zjacobi", GAMMA), v0=problems.rational_complex_start_vector(N), **kw)
        out["cases"][name] = dict(ret=r.ret, initSize=r.initSize, evals=np.asarray(r.evals).tolist(), resNorms=np.asarray(r.resNorms).tolist(),
                                  stats={k: r.stats[k] for k in ("numOuterIterations", "numMatvecs", "numRestarts", "numPreconds")})
        print(name, r.ret, r.initSize, out["cases"][name]["stats"], r.evals)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_zprecond.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
