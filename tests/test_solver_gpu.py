"""End-to-end parity of the HIP path (hip_dprimme / hip_sprimme through the C ABI of
primme_amd/libprimme_amd.so) on a real MI355X:
  - against the committed reference fixtures (outputs of the real reference dprimme),
  - against the oracle (product host solver over the plain-C kernel restatement) on the same
    seeded inputs,
  - at BASELINE.json's full size (configs[1], n = 2 000 250) through size-independent
    properties: analytic spectrum, residual threshold, recomputed true residuals,
    orthonormality (the reference's own check_solution, tests/COMMON/ioandtest.c:96-145).
Tolerance (north star): eigenvalues 1e-10 relative to |A| in double, 1e-4 in float."""
import json
import os

import numpy as np
import pytest

from primme_amd import problems
from checkers import eigsh, Operator
from primme_amd import _ffi as F

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_solves.json")))
EXACT_HISTORY = []


def _make_v0(spec, n):
    if spec is None:
        return None
    if spec == "start_vector":
        return problems.start_vector(n)
    return np.random.default_rng(spec["rng"]).standard_normal((n, 9))[:, :spec["cols"]]

LOOSE = {"lap2d_closest_abs": 0.05, "lap2d_closest_geq": 0.05, "lap2d_closest_leq": 0.05,
         "ref_closest_abs": 0.25, "ref_closest_geq": 0.25, "ref_closest_leq_jdqmr": 0.25, "ref_soft": 0.25, "ref_two_shifts": 0.25,
         "harm_closest_abs": 0.25, "harm_closest_geq": 0.25, "harm_closest_leq_jdqmr": 0.25, "harm_two_shifts": 0.25,
         "jdqmr_blk4": 0.3, "jdqmr_etol_blk8_jacobi": 0.3, "jdqmr_closest_abs": 0.3}
# Block JDQMR is a different (equally valid) block iteration from the reference's, which indexes some QMR recurrences by
# block position and others by original column; unpreconditioned interior runs are chaotic at any
# block size.  profiles/r03_jdqmr_block_count_sweep.txt (48 random configurations against the live reference): operator
# applications within 0.79-1.18 of dprimme's (median 1.01), outer iterations 0.58-1.10 (fewer, longer inner solves at
# b = 4, 8), block size 1 exact.  The work measure (matvecs) gets the tight bound, the outer count the loose one.
LOOSE_MATVECS = {"jdqmr_blk4": 0.15, "jdqmr_etol_blk8_jacobi": 0.15, "jdqmr_closest_abs": 0.15}
# PRIMME_DYNAMIC switches between GD+k and JDQMR on WALL-CLOCK ratios (reference src/eigs/main_iter.c:2196-2407; SURVEY
# section 2 #4 marks it out of scope): neither its history nor the reference's is reproducible, so no count is compared.
# What must hold: it converges to the reference's pairs, it ends in one of the states a finished dynamic run can be in
# (dynamicMethodSwitch -1 = closing with JDQMR, -2 = GD+k, -3 = GD+k for few pairs), and the work stays within the band the
# two methods span (the fixtures' own counts are one draw from it).
DYNAMIC = {"lap1d_ex_dseq_dynamic", "lap3d_dynamic", "lap2d_dynamic_few_soft"}


def _case(name):
    g = GOLD[name]
    dims = tuple(g["dims"])
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(g["kwargs"])
    kw["v0"] = _make_v0(kw.get("v0"), n)
    if "dtype" in kw:
        kw["dtype"] = np.dtype(kw["dtype"])
    return op, kw, g


def test_product_library_is_the_one_loaded(built):
    lib = F.load_product()
    maps = open("/proc/self/maps").read()
    assert "primme_amd/libprimme_amd.so" in maps
    # the product library links neither the oracle nor the reference build
    deps = os.popen("ldd " + F.PRODUCT_LIB).read()
    assert "hostcheck" not in deps and "primme_ref" not in deps and "hipk_cpu" not in deps


@pytest.mark.parametrize("name", sorted(GOLD))
def test_hip_against_reference_fixture(built, name):
    op, kw, g = _case(name)
    r = eigsh(op, backend="hip", **kw)
    aN = g["params"]["aNorm"] if g["params"]["aNorm"] > 0 else max(abs(np.array(g["evals"])))
    assert r.ret == 0 and r.initSize == g["initSize"]
    ev, evg = np.array(r.evals), np.array(g["evals"])
    if name in ("lap2d_closest_abs", "jdqmr_closest_abs") or name.startswith("harm_") or name.startswith("ref_"):
        ev, evg = np.sort(ev), np.sort(evg)
    rel = 1e-4 if str(g["kwargs"].get("dtype", "")) == "float32" else 1e-10
    assert np.max(np.abs(ev - evg)) <= rel * aN
    thr = (g["kwargs"].get("eps") or 0) * aN
    if thr > 0:
        assert np.all(r.resNorms <= thr * (1 + 1e-6))
    its, itsg = r.stats["numOuterIterations"], g["stats"]["numOuterIterations"]
    if name in DYNAMIC:
        assert r.params["dynamicMethodSwitch"] in (-1, -2, -3), r.params["dynamicMethodSwitch"]
        assert g["stats"]["numMatvecs"] / 4 <= r.stats["numMatvecs"] <= 4 * g["stats"]["numMatvecs"]
        return
    # the device reductions add in a different order than the CPU BLAS: counts agree closely,
    # exactly for most cases; allow 2 % (5 % for the interior targets, where the reference
    # itself varies from run to run)
    assert abs(its - itsg) <= max(2, LOOSE.get(name, 0.02) * itsg), (its, itsg)
    if name in LOOSE_MATVECS:
        assert abs(r.stats["numMatvecs"] - g["stats"]["numMatvecs"]) <= LOOSE_MATVECS[name] * g["stats"]["numMatvecs"]
    if name not in LOOSE and its == itsg and r.stats["numMatvecs"] == g["stats"]["numMatvecs"]:
        # same convergence history as the reference: its residual norms are reproduced too (north
        # star: eigenvalues AND residual norms within 1e-10 |A| in double, 1e-4 |A| in float)
        assert np.max(np.abs(np.array(r.resNorms, dtype=np.float64) - np.array(g["resNorms"]))) <= rel * aN
        EXACT_HISTORY.append(name)


def test_most_fixtures_reproduce_the_reference_history(built):
    """The residual-norm comparison above only bites when the iteration / matvec counts are the
    reference's: make sure that is the rule, not the exception (runs after the fixture cases)."""
    print("exact history on the device:", len(EXACT_HISTORY), "of", len(GOLD), sorted(EXACT_HISTORY))      # (pytest -s / -rP shows it; no file is written)
    # Round 5, measured on the MI355X (gpurun_out/exact_history_gpu.json, twice, with the row-pattern SpMV and the iteration
    # enqueued ahead of the host both on): these 32 of the 50 fixtures reproduce the reference's outer-iteration AND matvec
    # counts exactly, and then its residual norms to 1e-10 |A| (1e-4 |A| in single precision) — every extremal-target
    # fixture (GD, GD+k, Olsen, LOBPCG-like, JDQMR at block size 1, blocks of 2 / 4 / 8 of the Davidson family, locking and
    # soft locking, both precisions).  The other 18 are the ones the tolerances above exist for: interior targets, harmonic /
    # refined extraction, block JDQMR, the wall-clock-driven dynamic method.  (lap3d_gdk sits on an edge: 465 iterations in the
    # reference, 464 or 465 here depending on the order in which the fused product sums t'At — with the row-pattern kernel in pairs
    # of rows it is 464: the two misses the assertion allows are for such cases.)
    known_exact = {"blk1_explicit", "blk2_implicit", "blk2_jacobi", "blk2_lock", "blk2_soft", "blk4_largest", "blk4_lock", "blk8_K40", "float_bs1",
                   "float_bs2", "jdqmr_bs1", "jdqmr_etol_bs1", "jdqmr_etol_jacobi", "jdqmr_etol_largest_3d", "jdqmr_float", "jdqmr_jacobi",
                   "jdqmr_largest_3d", "jdqmr_soft", "lap1d_ex_dseq", "lap2d_gd", "lap2d_gd_olsen", "lap2d_gdk_lock", "lap2d_gdk_soft",
                   "lap2d_gdk_soft_K20", "lap2d_jacobi", "lap2d_krylov_rng", "lap2d_largest", "lap3d_gdk", "lap3d_lobpcg", "lap3d_medium",
                   "lap3d_noanorm", "lobpcg_default"}
    missing = sorted(known_exact - set(EXACT_HISTORY))
    assert len(EXACT_HISTORY) >= 30 and len(missing) <= 2, (sorted(EXACT_HISTORY), missing)


EVECS = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_evecs.npz"))
EVEC_NAMES = sorted({k.split("/")[0] for k in EVECS.files})


def subspace_check(name, r, g):
    """Angle between the invariant subspace a solve returned and the one the REAL reference returned for the same fixture
    (tests/golden/reference_evecs.npz, generated by tests/golden/make_evec_golden.py): a parity check that needs neither an
    identical history nor residual norms.  Both bases are accurate to |r| / gap, gap = distance of the wanted eigenvalues to
    the rest of the (analytic) spectrum, so sin(largest principal angle) <= 2 eps |A| / gap + rounding."""
    dims = tuple(g["dims"])
    aN = g["params"]["aNorm"] if g["params"]["aNorm"] > 0 else max(abs(np.array(g["evals"])))
    Xr = EVECS[name + "/evecs"]
    X = np.asarray(r.evecs, dtype=np.float64)
    assert X.shape == Xr.shape, (name, X.shape, Xr.shape)
    Qr, _ = np.linalg.qr(Xr)
    Q, _ = np.linalg.qr(X)
    sin_max = float(np.linalg.norm(Q - Qr @ (Qr.T @ Q), 2))
    n = int(np.prod(dims))
    spec = problems.laplacian_eigenvalues(dims, n)                    # the whole analytic spectrum, ascending
    wanted = np.sort(np.array(g["evals"], dtype=np.float64))
    idx = [int(np.argmin(np.abs(spec - w))) for w in wanted]
    taken = np.zeros(n, dtype=bool)
    for i, w in zip(idx, wanted):                                     # (a repeated eigenvalue: take the next free copy)
        j = i
        while taken[j] and j + 1 < n and abs(spec[j + 1] - w) <= 1e-9 * aN: j += 1
        taken[j] = True
    rest = spec[~taken]
    gap = float(np.min(np.abs(rest[None, :] - wanted[:, None])))
    eps = g["kwargs"].get("eps") or 1e-12
    assert gap > 1e-6 * aN, (name, gap)                               # the fixtures' wanted sets are separated from the rest
    tol = 2.0 * eps * aN / gap + 1e-12
    assert sin_max <= tol, (name, sin_max, tol, gap)
    # the eigenvalues themselves against the ANALYTIC spectrum with the bound the residual norms imply (Kato-Temple:
    # |theta - lambda| <= |r|^2 / gap_i, gap_i = distance to the nearest OTHER eigenvalue) — much tighter than the 1e-10 |A| of the
    # fixture comparison, and stated relative to the eigenvalue too (every other tolerance in this file is relative to |A|)
    lam = np.sort(np.asarray(r.evals, dtype=np.float64))
    for li, ji in zip(lam, sorted(np.flatnonzero(taken))):
        others = np.delete(spec, ji)
        gap_i = float(np.min(np.abs(others - spec[ji])))
        bound = (eps * aN) ** 2 / gap_i + 200 * np.finfo(np.float64).eps * aN if gap_i > 1e-6 * aN else eps * aN
        assert abs(li - spec[ji]) <= bound, (name, li, spec[ji], bound)
        assert abs(li - spec[ji]) / max(abs(spec[ji]), 1e-300) <= bound / max(abs(spec[ji]), 1e-300)
    return sin_max, tol


@pytest.mark.parametrize("name", EVEC_NAMES)
def test_hip_invariant_subspace_against_the_references_eigenvectors(built, name):
    """The 17 fixtures whose history is not reproduced count for count (interior targets, harmonic / refined extraction, block
    JDQMR, the dynamic method): the returned invariant subspace is the reference's to 2 eps |A| / gap."""
    op, kw, g = _case(name)
    r = eigsh(op, backend="hip", **kw)
    assert r.ret == 0 and r.initSize == g["initSize"]
    subspace_check(name, r, g)


def test_block_jdqmr_with_the_references_own_indexing_on_the_device(built, monkeypatch):
    """PRIMME_AMD_JDQMR_REF_INDEXING=1 (csrc/eigs_jd.c; tests/test_solver_host.py has the CPU-checker and live-reference legs):
    the block QMR recurrences indexed the way the reference indexes them.  On the HIP path the block fixture then follows
    dprimme's history — the same outer-iteration, matvec and restart counts, the same residual norms — where the default
    (every recurrence with its own column) is a different iteration covered only by the 15 % / 30 % tolerances above."""
    op, kw, g = _case("jdqmr_blk4")
    monkeypatch.setenv("PRIMME_AMD_JDQMR_REF_INDEXING", "1")
    r = eigsh(op, backend="hip", **kw)
    got = (r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"])
    want = (g["stats"]["numOuterIterations"], g["stats"]["numMatvecs"], g["stats"]["numRestarts"])
    print("jdqmr_blk4 with the reference's indexing on the device:", got, "reference:", want)
    assert r.ret == 0 and np.max(np.abs(np.array(r.evals) - np.array(g["evals"]))) <= 1e-10 * 8.0
    assert got == want, (got, want)
    assert np.max(np.abs(np.array(r.resNorms) - np.array(g["resNorms"]))) <= 1e-10 * 8.0
    op, kw, g = _case("jdqmr_etol_blk8_jacobi")
    r = eigsh(op, backend="hip", **kw)
    print("jdqmr_etol_blk8_jacobi:", (r.stats["numOuterIterations"], r.stats["numMatvecs"]), "reference:", g["stats"])
    assert abs(r.stats["numMatvecs"] - g["stats"]["numMatvecs"]) <= 0.03 * g["stats"]["numMatvecs"]
    assert abs(r.stats["numOuterIterations"] - g["stats"]["numOuterIterations"]) <= 4


@pytest.mark.parametrize("K,mr", [(260, 60), (520, 100), (1000, 150)])
def test_hip_wide_basis(built, K, mr):
    """maxBasisSize beyond 255 on the device (round 6: ritz_big_kernel with 16 / 8 rows per tile takes 511 / 1 023 columns):
    the restart through the wide-basis update, against the analytic spectrum and — for the two sizes the CPU checker solves in
    seconds — against the checker's history (tests/test_solver_host.py::test_wide_basis_against_live_reference ties that one
    to the live reference)."""
    dims = (50, 60)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=8, eps=1e-11, aNorm=8.0, maxBasisSize=K, minRestartSize=mr, v0=problems.start_vector(n), maxBlockSize=1)
    a = eigsh(op, backend="hip", **kw)
    assert a.ret == 0 and a.initSize == 8 and a.params["maxBasisSize"] == K
    assert np.max(np.abs(a.evals - problems.laplacian_eigenvalues(dims, 8))) <= 1e-10 * 8.0
    assert np.all(a.resNorms <= 1e-11 * 8.0 * (1 + 1e-6))
    AX = problems.csr_matvec_numpy(rp, ci, va, a.evecs)
    assert np.all(np.linalg.norm(AX - a.evecs * a.evals, axis=0) <= 2e-11 * 8.0 + 1e-13)
    if K == 260:
        b = eigsh(op, backend="hostcheck", **kw)
        assert a.stats["numRestarts"] == b.stats["numRestarts"] >= 2
        assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.03 * b.stats["numOuterIterations"]


@pytest.mark.parametrize("name", ["gen_gdk", "gen_gdk_blk2", "gen_largest_soft", "gen_olsen_jacobi", "gen_gd", "gen_lobpcg", "gen_blk4_3d", "gen_noanorm",
                                  "gen_lund_gdk", "gen_lund_jdqmr", "gen_lund_blk2"])
def test_hip_generalized_problem_against_reference_fixture(built, name):
    """Generalised problems A x = lambda B x on the device (round 6; tests/generalized_cases.py): the reference's eigenvalues and
    residual norms, scipy's dense truth, B-orthonormal vectors, counts within 5 % of dprimme's."""
    from generalized_cases import check
    check(name, "hip")


@pytest.mark.parametrize("name", ["gen_jdqmr", "gen_jdqmr_jacobi", "gen_jdqmr_etol_3d", "gen_jdqmr_largest", "gen_jdqmr_blk3", "gen_jd_olsen", "gen_jdqmr_soft"])
def test_hip_generalized_jdqmr_against_reference_fixture(built, name):
    """The JDQMR inner solver with a mass matrix on the device (round 6; csrc/eigs_jd.c, the CPU-checker leg with the exact counts is
    tests/test_solver_host.py): the reference's eigenvalues, scipy's dense truth, B-orthonormal vectors with true residuals, outer
    iterations within 15 % of dprimme's (inner-outer histories separate at rounding level between two arithmetic orders)."""
    from generalized_cases import check
    check(name, "hip")


@pytest.mark.parametrize("kw", [dict(method="GD_plusK"), dict(method="JDQMR", locking=1, precond="jacobi"), dict(method="GD_plusK", maxBlockSize=2)])
def test_hip_generalized_single_precision(built, kw):
    """hip_sprimme with a mass matrix on the device against scipy's dense truth and the CPU checker's counts (the checker is tied
    to live sprimme in tests/test_solver_host.py)."""
    import scipy.linalg as sl, scipy.sparse as sp
    from test_solver_host import _single_generalized
    a, b = _single_generalized("hip", kw), _single_generalized("hostcheck", kw)
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    w = sl.eigh(sp.csr_matrix((va, ci, rp), shape=(n, n)).toarray(), sp.csr_matrix((bva, bci, brp), shape=(n, n)).toarray(), eigvals_only=True)[:4]
    assert a.ret == b.ret == 0 and np.max(np.abs(a.evals - w)) <= 1e-4 * 8.0
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= max(2, 0.15 * b.stats["numOuterIterations"])


def test_hip_generalized_dynamic_method(built):
    """PRIMME_DYNAMIC (the default method) with a mass matrix on the device: both modes of the switch run with B."""
    import scipy.linalg as sl, scipy.sparse as sp
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", mass=Operator(n, csr=(brp, bci, bva)), numEvals=5, eps=1e-9, aNorm=8.0,
              v0=problems.start_vector(n), method="DYNAMIC")
    w = sl.eigh(sp.csr_matrix((va, ci, rp), shape=(n, n)).toarray(), sp.csr_matrix((bva, bci, brp), shape=(n, n)).toarray(), eigvals_only=True)[:5]
    assert r.ret == 0 and np.max(np.abs(r.evals - w)) <= 1e-9 * 8.0


def test_hip_refined_extraction_with_an_extremal_target(built):
    """Round 6 widening (VERDICT r05 Missing #4): refined extraction with target = largest / largest_abs and the shift of the
    factorisation given in targetShifts — accepted by the reference's check_input (primme_c.c:512-520), returned -44 here until
    now.  The CPU-checker leg against the LIVE reference is in tests/test_solver_host.py; here the HIP path against the
    checker: the same pairs (sorted: the two return locked pairs in their order of convergence), iteration counts within 10 %."""
    dims = (20, 21)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    for tgt, sh in (("largest", 8.0), ("largest_abs", 8.0)):
        kw = dict(numEvals=4, target=tgt, targetShifts=[sh], eps=1e-9, aNorm=8.0, projection="refined", v0=problems.start_vector(n))
        a = eigsh(op, backend="hip", **kw)
        b = eigsh(op, backend="hostcheck", **kw)
        assert a.ret == b.ret == 0 and a.initSize == b.initSize == 4
        assert np.max(np.abs(np.sort(a.evals) - np.sort(b.evals))) <= 1e-10 * 8.0
        assert np.all(a.resNorms <= 1e-9 * 8.0 * (1 + 1e-6))
        assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= max(3, 0.1 * b.stats["numOuterIterations"])


@pytest.mark.parametrize("dims,kw", [
    ((30, 31, 32), dict(numEvals=10, eps=1e-8, aNorm=12.0)),
    ((64, 63), dict(numEvals=6, eps=1e-9, aNorm=8.0, method="GD_Olsen_plusK")),
    ((40, 41, 39), dict(numEvals=5, eps=1e-8, aNorm=12.0, precond="jacobi", target="largest")),
    ((50, 51), dict(numEvals=4, eps=1e-9, aNorm=8.0, maxBlockSize=2, orth=F.primme_orth_implicit_I)),
])
def test_hip_against_oracle(built, dims, kw):
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    bs = kw.get("maxBlockSize", 1)
    rng = np.random.default_rng(1)
    v0 = problems.start_vector(n) if bs == 1 else rng.standard_normal((n, bs))
    a = eigsh(op, backend="hip", v0=v0, **kw)
    b = eigsh(op, backend="hostcheck", v0=v0, **kw)
    assert a.ret == b.ret == 0 and a.initSize == b.initSize
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * kw["aNorm"]
    assert np.all(a.resNorms <= kw["eps"] * kw["aNorm"])
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= max(2, 0.03 * b.stats["numOuterIterations"])
    # same invariant subspace: |V_hip' V_oracle| = identity up to signs
    G = np.abs(a.evecs.T @ b.evecs)
    assert np.max(np.abs(G - np.eye(G.shape[0]))) < 1e-5


def test_stencil_operator_equals_csr(built):
    dims = (33, 35, 31)
    rp, ci, va, n = problems.laplacian_csr(dims)
    v0 = problems.start_vector(n)
    a = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", numEvals=5, eps=1e-9, aNorm=12.0, v0=v0)
    b = eigsh(Operator(n, stencil=dims), backend="hip", numEvals=5, eps=1e-9, aNorm=12.0, v0=v0)
    assert a.ret == b.ret == 0
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-11 * 12
    assert a.stats["numMatvecs"] == b.stats["numMatvecs"]


def test_float_path(built):
    """hip_sprimme (orth forced to implicit_I: the explicit_I block path is a later row);
    tolerance 1e-4 relative (north star)."""
    dims = (24, 25)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=4, eps=1e-4, aNorm=8.0, v0=problems.start_vector(n), orth=F.primme_orth_implicit_I)
    a = eigsh(op, backend="hip", dtype=np.float32, **kw)
    b = eigsh(op, backend="hostcheck", dtype=np.float32, **kw)
    exact = problems.laplacian_eigenvalues(dims, 4)
    assert a.ret == b.ret == 0
    assert np.max(np.abs(a.evals - exact)) <= 1e-4 * 8.0
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-4 * 8.0


def test_unsupported_configurations_fail_loudly(built):
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    op = Operator(n, csr=(rp, ci, va))
    rpw, ciw, vaw, nw = problems.laplacian_csr((40, 41))
    r = eigsh(Operator(nw, csr=(rpw, ciw, vaw)), backend="hip", numEvals=2, maxBasisSize=1100, aNorm=8.0, v0=problems.start_vector(nw))
    assert r.ret == -44      # PRIMME_FUNCTION_UNAVAILABLE (a basis beyond 1 023 columns), no silent CPU fallback
    # host (non-device) evecs pointer is rejected like the reference's GPU flavour does (-31)
    import ctypes as C
    lib = F.load_product()
    p = F.PrimmeParams()
    lib.primme_initialize(C.byref(p))
    p.n = 50; p.numEvals = 2; p.matrixMatvec = 1
    lib.primme_set_method(F.PRIMME_GD_plusK, C.byref(p))
    ev = np.zeros(2); rn = np.zeros(2); vec = np.zeros((2, 50))
    assert lib.hip_dprimme(ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -31


def test_full_size_config2_properties(built):
    """BASELINE.json configs[1] at full size: 3-D 7-pt Laplacian 125x126x127, 10 smallest."""
    dims = (125, 126, 127)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    aN, eps, nev = 12.0, 1e-8, 10
    r = eigsh(op, backend="hip", numEvals=nev, eps=eps, aNorm=aN, v0=problems.start_vector(n))
    assert r.ret == 0 and r.initSize == nev
    exact = problems.laplacian_eigenvalues(dims, nev)
    assert np.max(np.abs(r.evals - exact)) <= 1e-10 * aN          # parity bar vs the analytic spectrum
    assert np.all(np.diff(r.evals) > 0)                            # returned sorted (locked values insertion-sorted)
    assert np.all(r.resNorms <= eps * aN)
    V = r.evecs
    assert np.max(np.abs(V.T @ V - np.eye(nev))) < 1e-7            # check_solution (i)
    AV = problems.csr_matvec_numpy(rp, ci, va, V)
    for i in range(nev):
        true_rn = np.linalg.norm(AV[:, i] - r.evals[i] * V[:, i])
        assert abs(V[:, i] @ AV[:, i] - r.evals[i]) <= max(r.resNorms[i], aN * 1e-14)   # (ii)
        assert true_rn <= eps * aN * 1.05 and abs(true_rn - r.resNorms[i]) <= 2 * true_rn + 1e-13   # (iii)


# ---- the reference's own driver regression cases on LUNDA.mtx (tests/tests/test_00N) --------
import reference_driver_cases as RD


@pytest.mark.parametrize("name", sorted(RD.CASES))
def test_hip_reference_driver_case(built, name):
    """HIP path on the reference's regression inputs, accepted by the reference driver's
    check_solution against the reference's stored eigenvectors (sol_00N_double), and equal to the
    oracle's eigenvalues."""
    rp, ci, va, n = RD.lunda()
    op = Operator(n, csr=(rp, ci, va))
    case = RD.CASES[name]
    r = eigsh(op, backend="hip", **case["kw"])
    assert r.ret == 0 and r.initSize == case["kw"]["numEvals"]
    X = RD.read_sol(case["sol"], n)
    bad = RD.check_solution(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                            r.evals, np.asarray(r.evecs, dtype=np.float64), r.resNorms, r.params["aNorm"],
                            case["kw"]["eps"], X)
    assert not bad, bad
    h = eigsh(op, backend="hostcheck", **case["kw"])
    assert h.ret == 0
    assert np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) <= 1e-10 * r.params["aNorm"]


def test_hip_tiled_lunda_block_jdqmr(built):
    """BASELINE configs[2] in small: block-diagonal tiles of LUNDA.mtx (tile t scaled by
    1 + t/1000, so the spectrum stays simple), JDQMR, block size 8, 20 eigenvalues closest to a
    shift; truth = union of the scaled dense spectra."""
    import ingest_c
    lib = F.load_product()
    rp, ci, va, n0, _ = ingest_c.mm_read(lib, os.path.join(RD.DATA, "LUNDA.mtx"))      # the C reader and tiler (row f3)
    T = 64
    scale = lambda t: 1.0 + t / 1000.0
    trp, tci, tva = ingest_c.tile_block_diagonal(lib, rp, ci, va, T, 1.0, 1.0 / 1000.0)
    ptrp, ptci, ptva = problems.tile_block_diagonal(rp, ci, va, T, scale)
    assert np.array_equal(trp, ptrp) and np.array_equal(tci, ptci) and np.allclose(tva, ptva, rtol=1e-15)
    n = n0 * T
    A = np.zeros((n0, n0))
    A[np.repeat(np.arange(n0), np.diff(rp)), ci] = va
    w0 = np.linalg.eigvalsh(A)
    w = np.concatenate([w0 * scale(t) for t in range(T)])
    shift = 1.0e7
    want = w[np.argsort(np.abs(w - shift))][:20]
    op = Operator(n, csr=(trp, tci, tva))
    r = eigsh(op, backend="hip", numEvals=20, target="closest_abs", targetShifts=[shift], method="JDQMR",
              maxBlockSize=8, eps=1e-10, aNorm=float(np.abs(w).max()), precond="jacobi")
    assert r.ret == 0
    assert np.max(np.abs(np.sort(r.evals) - np.sort(want))) <= 1e-10 * np.abs(w).max()
    assert np.all(r.resNorms <= 1e-10 * np.abs(w).max() * (1 + 1e-6))


@pytest.mark.parametrize("kw", [dict(numEvals=6, method="JDQR"), dict(numEvals=6, method="JDQR", locking=0),
                                dict(numEvals=4, method="JD_Olsen_plusK")])
def test_hip_skew_projectors_and_exact_olsen(built, kw):
    """K^-1-weighted projectors (JDQR) and the exact Olsen correction on the device against the oracle."""
    rp, ci, va, n = RD.lunda()
    op = Operator(n, csr=(rp, ci, va))
    args = dict(eps=1e-10, target="largest", precond=("jacobi", 3e8), **kw)
    r = eigsh(op, backend="hip", **args)
    h = eigsh(op, backend="hostcheck", **args)
    assert r.ret == 0 and h.ret == 0
    assert np.max(np.abs(np.sort(r.evals) - np.sort(h.evals))) <= 1e-10 * r.params["aNorm"]
    assert np.all(r.resNorms <= 1e-10 * r.params["aNorm"] * (1 + 1e-6))
    assert abs(r.stats["numOuterIterations"] - h.stats["numOuterIterations"]) <= max(3, 0.1 * h.stats["numOuterIterations"])


def test_hip_device_rayleigh_ritz_option(built, monkeypatch):
    """PRIMME_AMD_DEVICE_RR: the small projected eigenproblem solved by the device Jacobi kernel
    instead of the host QL solver; same eigenpairs."""
    dims = (30, 31, 32)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=6, eps=1e-9, aNorm=12.0, v0=problems.start_vector(n))
    a = eigsh(op, backend="hip", **kw)
    monkeypatch.setenv("PRIMME_AMD_DEVICE_RR", "1")
    b = eigsh(op, backend="hip", **kw)
    assert a.ret == 0 and b.ret == 0
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * 12.0
    assert np.max(np.abs(b.evals - problems.laplacian_eigenvalues(dims, 6))) <= 1e-10 * 12.0
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.05 * a.stats["numOuterIterations"] + 2


@pytest.mark.parametrize("projection", ["RR", "harmonic", "refined"])
def test_hip_interior_pairs_medium_size(built, projection):
    """Interior eigenvalues of a 3-D Laplacian (analytic spectrum) with the three extractions, JDQMR
    as the correction: right values, residuals under the threshold, orthonormal."""
    dims = (20, 21, 19)
    rp, ci, va, n = problems.laplacian_csr(dims)
    i, j, k = np.meshgrid(*(np.arange(1, d + 1) for d in dims), indexing="ij")
    w = (6 - 2 * np.cos(i * np.pi / (dims[0] + 1)) - 2 * np.cos(j * np.pi / (dims[1] + 1)) - 2 * np.cos(k * np.pi / (dims[2] + 1))).ravel()
    shift = 1.0
    want = np.sort(w[np.argsort(np.abs(w - shift))][:4])
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", numEvals=4, target="closest_abs", targetShifts=[shift], eps=1e-8,
              aNorm=12.0, method="JDQMR", projection=projection, v0=problems.start_vector(n), maxMatvecs=200000)
    assert r.ret == 0
    assert np.max(np.abs(np.sort(r.evals) - want)) <= 1e-8 * 12.0
    assert np.all(r.resNorms <= 1e-8 * 12.0 * (1 + 1e-6))
    X = np.asarray(r.evecs, dtype=np.float64)
    assert np.linalg.norm(X.T @ X - np.eye(4)) <= 1e-7


def test_hip_application_matvec_callback_keeps_the_restart_paths(built):
    """An application's own device callback instead of primme_amd_matvec (the drop-in case of
    examples/ex_eigs_dhipblas.c): the iteration tail is project / scale / callback / t'At instead of the one-launch
    form, the fused and speculative restart still apply; iteration, matvec and restart counts are the reference's."""
    import ctypes as C
    from checkers import Session
    from primme_amd import _ffi as F
    dims = (40, 41)
    rp, ci, va, n = problems.laplacian_csr(dims)
    s = Session(Operator(n, csr=(rp, ci, va)), backend="hip")
    A = [h for k, h in s.handles if k == "csr"][0]
    calls = [0]

    def matvec(x, ldx, y, ldy, bs, pp, ierr):
        calls[0] += bs[0]
        stream = C.cast(pp[0].queue, C.POINTER(C.c_void_p))[0]      # the solver's stream (primme->queue), like the reference's handle
        ierr[0] = s.lib.hipk_csr_matvec(A, stream, x, ldx[0], y, ldy[0], bs[0])
    cb = F.BLOCK_OP(matvec)
    kw = dict(numEvals=10, eps=1e-10, aNorm=8.0, v0=problems.start_vector(n), method="GD_plusK")
    r = s.solve(user_matvec=cb, **kw)
    own = s.solve(**kw)
    s.close()
    h = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", **kw)
    assert r.ret == 0 and own.ret == 0 and h.ret == 0
    assert np.max(np.abs(r.evals - problems.laplacian_eigenvalues(dims, 10))) <= 1e-10 * 8.0
    for key in ("numOuterIterations", "numMatvecs", "numRestarts"):
        assert abs(r.stats[key] - h.stats[key]) <= max(2, 0.02 * h.stats[key]), key
        assert abs(own.stats[key] - h.stats[key]) <= max(2, 0.02 * h.stats[key]), key
    # (speculative applications that were discarded are not counted by the solver)
    assert r.stats["numMatvecs"] <= calls[0] <= r.stats["numMatvecs"] + 30
