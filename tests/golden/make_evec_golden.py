"""Eigenvectors of the REAL reference for the fixtures whose convergence history is NOT reproduced count for count (interior
targets, harmonic / refined extraction, block JDQMR, the wall-clock-driven dynamic method): tests/golden/reference_evecs.npz.
With them the parity tests have a check that does not depend on the history or on residual norms — the angle between the
returned invariant subspace and the reference's (tests/test_solver_gpu.py, tests/test_solver_host.py).
Run in the build container only (needs oracle/_ref):  python tests/golden/make_evec_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from primme_amd import problems  # noqa: E402
from checkers import eigsh, Operator  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "reference_solves.json")))
# every fixture the solver tests give a count tolerance to (tests/test_solver_gpu.py: LOOSE, DYNAMIC)
NAMES = ["lap2d_closest_abs", "lap2d_closest_geq", "lap2d_closest_leq", "ref_closest_abs", "ref_closest_geq", "ref_closest_leq_jdqmr",
         "ref_soft", "ref_two_shifts", "harm_closest_abs", "harm_closest_geq", "harm_closest_leq_jdqmr", "harm_two_shifts",
         "jdqmr_blk4", "jdqmr_etol_blk8_jacobi", "jdqmr_closest_abs", "lap1d_ex_dseq_dynamic", "lap2d_dynamic_few_soft"]
# (lap3d_dynamic, the eighteenth, is left out: 29 760 x 12 doubles would triple the file; the dynamic method is covered by the other two)


def make_v0(spec, n):
    if spec is None:
        return None
    if spec == "start_vector":
        return problems.start_vector(n)
    return np.random.default_rng(spec["rng"]).standard_normal((n, 9))[:, :spec["cols"]]


def main():
    out = {}
    for name in NAMES:
        g = GOLD[name]
        dims = tuple(g["dims"])
        rp, ci, va, n = problems.laplacian_csr(dims)
        kw = dict(g["kwargs"])
        kw["v0"] = make_v0(kw.get("v0"), n)
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", **kw)
        assert r.ret == 0 and r.initSize == g["initSize"], name
        X = np.asarray(r.evecs, dtype=np.float64)
        # the run must have arrived at the fixture's eigenvalues (the dynamic method's path is timing dependent, its result is not)
        assert np.max(np.abs(np.sort(r.evals) - np.sort(np.array(g["evals"])))) <= 1e-9 * max(1.0, abs(np.array(g["evals"])).max()), name
        res = np.linalg.norm(problems.csr_matvec_numpy(rp, ci, va, X) - X * np.asarray(r.evals), axis=0)
        out[name + "/evecs"] = X
        out[name + "/evals"] = np.asarray(r.evals, dtype=np.float64)
        print(name, X.shape, "largest true residual", res.max())
    np.savez_compressed(os.path.join(HERE, "reference_evecs.npz"), **out)


if __name__ == "__main__":
    main()
