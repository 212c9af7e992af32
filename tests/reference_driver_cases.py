"""The reference's own regression cases on LUNDA.mtx (tests/tests/test_001..007) and the
acceptance test its driver applies to every solve (check_solution, tests/COMMON/ioandtest.c:71-157),
restated for use by the CPU (hostcheck / reference) and GPU (hip) test modules.

Data under tests/golden/reference_driver/ are the reference's test DATA files: the Matrix-Market
matrix and the stored eigenvector files sol_00N_double (binary: [sizeof(scalar), n, cols], then
cols columns of n doubles, then a parameter-block dump that is ignored here, cf.
ioandtest.c:159-206)."""
import os
import numpy as np

from primme_amd import problems

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "reference_driver")

# primme.* settings of tests/tests/test_00N (driver.* lines become precond / sol); aNorm is left at
# 0 (the driver only overrides a negative aNorm, tests/driver.c:575), initial guesses are PRIMME's own
CASES = {
    # test_001: "unrestarted configuration"
    "test_001": dict(sol="sol_001_double", kw=dict(numEvals=5, eps=1e-12, maxBasisSize=140, minRestartSize=1,
                     maxBlockSize=1, maxMatvecs=140, target="largest", locking=1, method="GD_Olsen_plusK")),
    # test_002: tiny basis
    "test_002": dict(sol="sol_002_double", kw=dict(numEvals=30, eps=1e-12, maxBasisSize=3, minRestartSize=1,
                     maxBlockSize=1, maxOuterIterations=7800, target="largest", locking=1, maxPrevRetain=1,
                     method="GD_Olsen_plusK")),
    # test_003: default values
    "test_003": dict(sol="sol_003_double", kw=dict(numEvals=50, eps=1e-12, maxOuterIterations=7500, target="largest",
                     method="GD_Olsen_plusK")),
    # test_004: interior, closest to 0
    "test_004": dict(sol="sol_004_double", kw=dict(numEvals=50, eps=1e-12, maxOuterIterations=7500,
                     target="closest_abs", targetShifts=[0.0], method="GD_Olsen_plusK")),
    # test_005: same with the driver's Jacobi preconditioner, shift 0
    "test_005": dict(sol="sol_005_double", kw=dict(numEvals=50, eps=1e-12, maxOuterIterations=7500,
                     target="closest_abs", targetShifts=[0.0], method="GD_Olsen_plusK", precond=("jacobi", 0.0))),
    # test_006: JDQMR with preconditioner on an extreme problem
    "test_006": dict(sol="sol_006_double", kw=dict(numEvals=5, eps=1e-12, maxBasisSize=50, minRestartSize=30,
                     maxOuterIterations=9000, target="largest", method="DEFAULT_MIN_TIME", precond=("jacobi", 3e8))),
    # test_004 with the refined extraction (same wanted pairs, same stored vectors)
    "test_004_refined": dict(sol="sol_004_double", kw=dict(numEvals=50, eps=1e-12, maxOuterIterations=7500,
                     target="closest_abs", targetShifts=[0.0], method="GD_Olsen_plusK", projection="refined")),
    # test_007: interior pairs through the harmonic extraction
    "test_007": dict(sol="sol_007_double", kw=dict(numEvals=50, eps=1e-12, maxOuterIterations=7500,
                     target="closest_abs", targetShifts=[0.0], method="GD_Olsen_plusK", projection="harmonic")),
}


# tests/tests/test_10N: the complex Hermitian cases on mhd1280b.mtx (stored vectors sol_10N_doublecomplex).
# hip_zprimme works on the real-equivalent form and needs about twice the operator applications of
# zprimme, so test_101's cap of 140 (its "unrestarted" point: maxBasisSize = maxMatvecs) is lifted;
# everything else is the file's setting.  test_102 / test_103 (27 / 50 pairs) are left to the real cases.
CASES_Z = {
    "test_101": dict(sol="sol_101_doublecomplex", kw=dict(numEvals=5, eps=1e-12, maxBasisSize=140, minRestartSize=1,
                     maxBlockSize=1, target="largest", locking=1, method="GD_Olsen_plusK")),
    "test_104": dict(sol="sol_104_doublecomplex", kw=dict(numEvals=2, eps=1e-8, maxMatvecs=100000, target="closest_abs",
                     targetShifts=[1e-2], method="DEFAULT_MIN_TIME")),
    # (the real form needs ~9000 outer iterations where zprimme needs ~3300: the file's cap of 4000 is lifted)
    "test_105": dict(sol="sol_105_doublecomplex", kw=dict(numEvals=2, eps=1e-8, maxOuterIterations=40000, target="closest_abs",
                     targetShifts=[1e-2], method="GD_Olsen_plusK", precond=("jacobi", -1e-2))),
    "test_106": dict(sol="sol_106_doublecomplex", kw=dict(numEvals=5, eps=1e-12, maxOuterIterations=200, target="largest",
                     method="DEFAULT_MIN_TIME", precond=("jacobi", 3e8))),
}


def mhd():
    rp, ci, va, n, _ = problems.read_matrix_market(os.path.join(DATA, "mhd1280b.mtx"))
    return rp, ci, va, n


def read_sol_z(name, n):
    d = np.fromfile(os.path.join(DATA, name), dtype=np.complex128)
    assert int(round(-d[0].real)) == 16 and int(d[1].real) == n
    cols = int(d[2].real)
    return d[3:3 + n * cols].reshape(cols, n).T.copy()


def lunda():
    rp, ci, va, n, _ = problems.read_matrix_market(os.path.join(DATA, "LUNDA.mtx"))
    return rp, ci, va, n


def read_sol(name, n):
    d = np.fromfile(os.path.join(DATA, name), dtype=np.float64)
    assert int(d[0]) == 8 and int(d[1]) == n
    cols = int(d[2])
    return d[3:3 + n * cols].reshape(cols, n).T.copy()


def check_solution(A_apply, evals, evecs, rnorms, aNorm, eps, X):
    """ioandtest.c:71-157.  evecs n x k (columns), X the stored vectors.  Returns the list of
    violated checks (empty = pass)."""
    bad = []
    k = len(evals)
    delta = aNorm if aNorm > 0 else np.inf
    for i in range(1, k):
        delta = min(delta, abs(evals[i] - evals[i - 1]))
    meps = np.finfo(np.float64).eps
    for i in range(k):
        v = evecs[:, i]
        h = evecs[:, :i + 1].conj().T @ v
        if np.linalg.norm(h[:i]) > 1e-7:
            bad.append(f"ortho[{i}]={np.linalg.norm(h[:i]):.2e}")
        if abs(np.sqrt(abs(h[i])) - 1) > 1e-7:
            bad.append(f"norm[{i}]")
        Ax = A_apply(v)
        eval0 = np.vdot(v, Ax)
        if abs(evals[i] - eval0) > max(rnorms[i], aNorm * eps):
            bad.append(f"rayleigh[{i}]={abs(evals[i] - eval0):.2e}")
        r = Ax - evals[i] * v
        rnorm0 = np.linalg.norm(r)
        if abs(rnorms[i] - rnorm0) > max(2 * rnorm0, 10 * max(aNorm, abs(evals[i])) * meps):
            bad.append(f"resnorm[{i}] {rnorms[i]:.2e} vs {rnorm0:.2e}")
        # residual after projecting out the returned vectors (one Gram-Schmidt pass)
        rp = r - evecs @ (evecs.conj().T @ r)
        if aNorm > 0 and np.linalg.norm(rp) > eps * aNorm * 2:
            bad.append(f"rr_residual[{i}]={np.linalg.norm(rp):.2e}")
        # angle against the stored invariant subspace
        prod = float(np.sum(np.abs(X.conj().T @ v) ** 2))
        bound = aNorm * eps / delta
        s2 = np.sqrt(2.0)
        if (s2 * prod + 1.0) / (s2 * bound + 1.0) < (s2 * prod - 1.0) / (1.0 - s2 * bound):
            bad.append(f"angle[{i}] cos={prod:.3e}")
    return bad


# ---- singular value cases (tests/tests/test_20N, driver tests/driversvds.c) -----------------
# The driver's default method is the hybrid one (normal equations, then the augmented operator).
SVDS_CASES = {
    "test_201": dict(sol="sol_201svds_double", kw=dict(numSvals=5, eps=1e-6, target="largest")),
    "test_202": dict(sol="sol_202svds_double", kw=dict(numSvals=5, eps=1e-12, target="largest")),
    # test_201 / test_202 as the driver runs them: the default (hybrid) method
    "test_201_hybrid": dict(sol="sol_201svds_double", kw=dict(numSvals=5, eps=1e-6, target="largest", method="default")),
    "test_202_hybrid": dict(sol="sol_202svds_double", kw=dict(numSvals=5, eps=1e-12, target="largest", method="default")),
    # test_207: the augmented operator alone
    "test_207": dict(sol="sol_207svds_double", kw=dict(numSvals=5, eps=1e-6, target="largest", method="augmented")),
    # test_205 / test_206: the smallest triplet with the driver's diagonal preconditioner (PrecChoice = jacobi:
    # 1 / diag(A'A), 1 / diag(AA'), tests/COMMON/mat.c:353-426), the driver's default hybrid method
    "test_205": dict(sol="sol_205svds_double", matrix="lund_b.mtx",
                     kw=dict(numSvals=1, eps=1e-12, target="smallest", method="default", precond="jacobi", methodStage1="DEFAULT_METHOD")),
    "test_206": dict(sol="sol_206svds_double", matrix="rect.mtx",
                     kw=dict(numSvals=1, eps=1e-12, target="smallest", method="default", precond="jacobi", methodStage1="DEFAULT_METHOD")),
}
# test_203 / test_204 (5 smallest triplets of lund_b.mtx / rect.mtx, sigma_min ~ 1e-9 |A|, eps 7e-12,
# no preconditioner) need more than 1e5 operator applications in the reference itself (it returns
# -103 / -203 under that cap); the smallest-triplet path (hybrid with the refined extraction in the
# augmented stage) is covered on well conditioned matrices in tests/test_svds_host.py instead.


def svds_matrix(name="rect.mtx"):
    rp, ci, va, m, n = problems.read_matrix_market(os.path.join(DATA, name))
    return rp, ci, va, m, n


def rect():
    rp, ci, va, m, n = problems.read_matrix_market(os.path.join(DATA, "rect.mtx"))
    return rp, ci, va, m, n


def read_sol_svds(name, m, n):
    d = np.fromfile(os.path.join(DATA, name), dtype=np.float64)
    assert int(d[0]) == 8 and int(d[1]) == m and int(d[2]) == n
    cols = int(d[3])
    U = d[4:4 + m * cols].reshape(cols, m).T.copy()
    V = d[4 + m * cols:4 + (m + n) * cols].reshape(cols, n).T.copy()
    return U, V


def check_solution_svds(A_apply, At_apply, svals, U, V, rnorms, aNorm, eps, XU):
    """ioandtest.c:266-345: unit vectors, u'Av = sigma, true residual of the triplet, angle of u
    against the stored left singular vectors."""
    bad = []
    k = len(svals)
    delta = aNorm
    for i in range(1, k):
        delta = min(delta, abs(svals[i] - svals[i - 1]))
    for i in range(k):
        u, v = U[:, i], V[:, i]
        if abs(1.0 - u @ u) > 1e-8: bad.append(f"normU[{i}]")
        if abs(1.0 - v @ v) > 1e-8: bad.append(f"normV[{i}]")
        Av = A_apply(v)
        s0 = u @ Av
        if abs(svals[i] - s0) > max(rnorms[i], aNorm * eps): bad.append(f"sval[{i}] {svals[i]} vs {s0}")
        r2 = np.sum((Av - svals[i] * u) ** 2) + np.sum((At_apply(u) - svals[i] * v) ** 2)
        rn0 = np.sqrt(r2)
        if rnorms[i] < rn0 and rn0 > 10 * rnorms[i]: bad.append(f"resnorm[{i}] {rnorms[i]:.2e} vs {rn0:.2e}")
        if rn0 > 8 * eps * aNorm * np.sqrt(i + 1.0): bad.append(f"rr_residual[{i}]={rn0:.2e}")
        prod = float(np.sum((XU.T @ u) ** 2))
        bound = aNorm * eps / delta
        s2 = np.sqrt(2.0)
        if (s2 * prod + 1.0) / (s2 * bound + 1.0) < (s2 * prod - 1.0) / (1.0 - s2 * bound):
            bad.append(f"angle[{i}] cos={prod:.3e}")
    return bad


# ---- the reference's interface tests (tests/Makefile:145-188, "tests_primme_interface") -------
# 1-D Laplacians of size 0..100, every preset method, numEvals up to n, four targets, RR and
# refined extraction: tiny and degenerate problem sizes, numEvals = n, bases that fill the space.
# The stored vectors are tests/tests/sol_testi-<n>-<numEvals>-<target>_double (copied as data
# under tests/golden/reference_driver/testi/).
TESTI_METHODS = ("DEFAULT_METHOD DYNAMIC DEFAULT_MIN_TIME DEFAULT_MIN_MATVECS Arnoldi GD_plusK GD_Olsen_plusK "
                 "JD_Olsen_plusK JDQR JDQMR JDQMR_ETol STEEPEST_DESCENT LOBPCG_OrthoBasis LOBPCG_OrthoBasis_Window").split()
TESTI_SIZES = (0, 1, 2, 3, 4, 5, 6, 7, 10, 100)


def testi_cases(method):
    """(n, numEvals, target, projection) exactly as the Makefile's nested loops and case filters."""
    for n in TESTI_SIZES:
        for nev in (0, 1, 2, 3, 4, 5, 6, 15, 100):
            if nev > n:
                continue
            for target in ("smallest", "largest", "closest_abs", "closest_geq"):
                if target == "closest_geq" and n in (4, 5, 6, 7) and nev == n:
                    continue
                if target == "closest_geq" and n == 100 and method.startswith("LOBPCG"):
                    continue
                closest = target.startswith("closest")
                for proj in ("RR", "refined"):
                    if closest and proj == "RR" and method.startswith("LOBPCG"):
                        continue
                    if closest and method.startswith(("STEEPEST_DESCENT", "Arnoldi", "GD")):
                        continue
                    if not closest and proj != "RR":
                        continue
                    yield n, nev, target, proj


def laplace1d(n):
    """laplace<n>.mtx of the Makefile (:190-196): tridiag(-1, 2, -1)."""
    rp, ci, va = [0], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, 2.0), (i + 1, -1.0)):
            if 0 <= j < n:
                ci.append(j); va.append(v)
        rp.append(len(ci))
    return np.array(rp, np.int32), np.array(ci, np.int32), np.array(va, np.float64)


def read_sol_testi(n, nev, target, cplx=False):
    d = np.fromfile(os.path.join(DATA, "testi", f"sol_testi-{n}-{nev}-primme_{target}_double" + ("complex" if cplx else "")),
                    dtype=np.complex128 if cplx else np.float64)
    cols = int(d[2].real)
    return d[3:3 + n * cols].reshape(cols, n).T.copy() if n * cols else np.zeros((n, 0), dtype=d.dtype)


def run_testi_case(eigsh, Operator, methods, backend, method, n, nev, target, proj, dtype=np.float64):
    """One interface case with the test files' settings (eps 1e-6, shift 0.5, maxMatvecs 50000);
    returns (ret, list of violated checks)."""
    rp, ci, va = laplace1d(n)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend=backend, numEvals=nev, eps=1e-6, target=target, targetShifts=[0.5],
              projection=proj, maxMatvecs=50000, method=methods.get(method, 0), dtype=dtype)
    if r.ret != 0 or nev == 0 or n == 0:
        return r.ret, []
    X = read_sol_testi(n, nev, target, cplx=np.dtype(dtype).kind == "c")
    k = r.initSize
    with np.errstate(divide="ignore"):
        bad = check_solution(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(), r.evals[:k],
                             np.asarray(r.evecs)[:, :k], r.resNorms[:k], r.params["aNorm"], 1e-6, X)
    if k < X.shape[1]:
        bad.append(f"returned {k} pairs, stored {X.shape[1]}")
    return r.ret, bad
