"""The reference's zprimme / cprimme outputs (tests/golden/reference_solves_complex.json, made by
tests/golden/make_complex_golden.py) and the one function that replays a case on a backend.  Shared by the CPU suite
(hostcheck: the product's complex host solver over the plain-C complex kernels) and the GPU suite (hip)."""
import json
import os

import numpy as np

from primme_amd import problems
from checkers import Operator, eigsh

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_solves_complex.json")))
# counts reproduce exactly in double precision; single precision (complex64) runs a different rounding in every
# kernel and may take a few iterations more or fewer
COUNT_KEYS = ("numOuterIterations", "numMatvecs", "numRestarts", "numPreconds")


def replay(name, backend, complex_form="native"):
    fx = FIX[name]
    n = fx["n"]
    kw = dict(fx["kwargs"])
    dtype = np.dtype(kw.pop("dtype", "complex128"))
    if "v0" not in kw:
        kw["v0"] = problems.complex_start_vector(n)
    if "iseed" in kw:
        kw["iseed"] = tuple(kw["iseed"])
    rp, ci, va = problems.hermitian_graded_csr(n)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend=backend, dtype=dtype, complex_form=complex_form, **kw)
    return fx, r, (rp, ci, va), dtype


def check(name, backend):
    fx, r, (rp, ci, va), dtype = replay(name, backend)
    single = dtype == np.complex64
    aN = fx["aNorm"]
    eps = fx["kwargs"]["eps"]
    assert r.ret == fx["ret"] == 0 and r.initSize == fx["initSize"], (name, r.ret, r.initSize)
    ev = np.asarray(r.evals, dtype=np.float64)
    assert np.max(np.abs(ev - np.array(fx["evals"]))) <= (1e-4 if single else 1e-10) * aN, name
    # (also the Davidson preconditioner on this graded diagonal: 1 / (d_j - theta) amplifies the last bits of theta, the two
    # histories separate by 1e-9 after ~30 iterations and end within 3 % of each other)
    interior = "closest" in fx["kwargs"].get("target", "") or "precond" in fx["kwargs"]
    refined = fx["kwargs"].get("projection") == "refined"
    for k in COUNT_KEYS:
        loose = 0.12 if ("precond" in fx["kwargs"] and "JDQMR" in fx["kwargs"].get("method", "")) else 0.03   # (inner iterations amplify it further)
        if fx["kwargs"].get("projection") == "harmonic" and "precond" not in fx["kwargs"]:
            loose = 0.10     # 1500 unpreconditioned interior iterations: 2 % on the CPU checker, 8 % on the GPU (different summation order)
        if refined:
            # the refined extraction on this unpreconditioned interior problem stagnates at a residual of 0.2 for a
            # hundred iterations before it locks on; the histories of two implementations separate at 1e-9 after 90
            # iterations (SVD of R by different algorithms) and the totals then differ like two different start vectors
            # would: +-30 % observed, for the real path against dprimme as well
            loose = 0.4
        if interior:
            # interior Ritz values move with the rounding of every inner product (a long run near the rounding floor of
            # the coefficient vectors): as for the real path the counts agree to within 3 %
            assert abs(r.stats[k] - fx["stats"][k]) <= loose * fx["stats"][k] + 1, (name, k, r.stats[k], fx["stats"][k])
        elif single:
            assert abs(r.stats[k] - fx["stats"][k]) <= 0.1 * fx["stats"][k] + 2, (name, k, r.stats[k], fx["stats"][k])
        else:
            assert r.stats[k] == fx["stats"][k], (name, k, r.stats[k], fx["stats"][k])
    if not single and not interior:
        assert np.max(np.abs(np.asarray(r.resNorms) - np.array(fx["resNorms"]))) <= 1e-10 * aN, name
    X = r.evecs.astype(np.complex128)
    AX = problems.csr_matvec_numpy(rp, ci, va, X)
    res = np.linalg.norm(AX - X * ev, axis=0)
    assert np.all(res <= 1.5 * eps * aN + 20 * np.finfo(np.float32 if single else np.float64).eps * aN), (name, res)
    assert np.max(np.abs(X.conj().T @ X - np.eye(X.shape[1]))) <= (1e-4 if single else 1e-9), name
    return r
