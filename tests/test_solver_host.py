"""Host control flow of the solver (the product's primme_amd/csrc/eigs_*.c) run over the plain-C
kernel layer (backend="hostcheck") against
  (a) the committed reference fixtures tests/golden/reference_solves.json, and
  (b) the real reference library when oracle/_ref/libprimme_ref.so is present.
Parity bar (BASELINE north star): eigenvalues within 1e-10 relative to |A|, every returned pair
below the residual threshold on both sides; iteration counts are compared too (equal for the
extremal GD+k runs, within a few % where the reference itself varies run to run)."""
import json
import os

import numpy as np
import pytest

from primme_amd import problems
from checkers import eigsh, Operator
from primme_amd import _ffi as F
import checkers

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_solves.json")))


def _make_v0(spec, n):
    if spec is None:
        return None
    if spec == "start_vector":
        return problems.start_vector(n)
    return np.random.default_rng(spec["rng"]).standard_normal((n, 9))[:, :spec["cols"]]

# cases whose iteration path is chaotic (interior targets: the reference itself differs run to run)
LOOSE = {"lap2d_closest_abs": 0.05, "lap2d_closest_geq": 0.05, "lap2d_closest_leq": 0.05,
         # block JDQMR: the reference's inner solver mixes position- and column-indexed scalars once a
         # block column has converged (see eigs_jd.c header); results agree, the paths do not
         "lap1d_ex_dseq_dynamic": 1e9, "lap3d_dynamic": 1e9, "lap2d_dynamic_few_soft": 1e9,   # timing-driven paths
         "ref_closest_abs": 0.1, "ref_closest_geq": 0.1, "ref_closest_leq_jdqmr": 0.15, "ref_soft": 0.1, "ref_two_shifts": 0.15,
         "harm_closest_abs": 0.1, "harm_closest_geq": 0.1, "harm_closest_leq_jdqmr": 0.15, "harm_two_shifts": 0.1,
         "jdqmr_blk4": 0.3, "jdqmr_etol_blk8_jacobi": 0.3, "jdqmr_closest_abs": 0.3}
# Block JDQMR is a different (equally valid) block iteration from the reference's, which indexes some QMR recurrences by
# block position and others by original column; unpreconditioned interior runs are chaotic at any
# block size.  profiles/r03_jdqmr_block_count_sweep.txt (48 random configurations against the live reference): operator
# applications within 0.79-1.18 of dprimme's (median 1.01), outer iterations 0.58-1.10 (fewer, longer inner solves at
# b = 4, 8), block size 1 exact.  The work measure (matvecs) gets the tight bound, the outer count the loose one.
LOOSE_MATVECS = {"jdqmr_blk4": 0.15, "jdqmr_etol_blk8_jacobi": 0.15, "jdqmr_closest_abs": 0.15}


def _run(name, backend):
    g = GOLD[name]
    dims = tuple(g["dims"])
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(g["kwargs"])
    kw["v0"] = _make_v0(kw.get("v0"), n)
    if "dtype" in kw:
        kw["dtype"] = np.dtype(kw["dtype"])
    return eigsh(op, backend=backend, **kw), g


@pytest.mark.parametrize("projection_column", ["from_Wtr", "pass_over_V"])
@pytest.mark.parametrize("name", sorted(GOLD))
def test_against_reference_fixture(built, name, projection_column, monkeypatch):
    """Two legs.  "from_Wtr" (the default): block size 1 GD forms the new column of H from W'r of the
    fused residual pass, H c and G = W'Q.  "pass_over_V" (PRIMME_AMD_NO_WTR=1): V'(A t)
    from a pass over V, as update_projection.c:99-122 computes it.  Both must reproduce the
    reference's iteration / matvec / restart counts."""
    if projection_column == "pass_over_V":
        monkeypatch.setenv("PRIMME_AMD_NO_WTR", "1")
    r, g = _run(name, "hostcheck")
    aN = g["params"]["aNorm"] if g["params"]["aNorm"] > 0 else max(abs(np.array(g["evals"])))
    assert r.ret == g["ret"] == 0
    assert r.initSize == g["initSize"]
    for k in ("maxBasisSize", "minRestartSize", "maxBlockSize", "locking", "orth", "maxPrevRetain"):
        assert r.params[k] == g["params"][k], k
    ev, evg = np.array(r.evals), np.array(g["evals"])
    if name in ("lap2d_closest_abs", "jdqmr_closest_abs") or name.startswith("harm_") or name.startswith("ref_"):
        ev, evg = np.sort(ev), np.sort(evg)
    rel = 1e-4 if str(g["kwargs"].get("dtype", "")) == "float32" else 1e-10
    assert np.max(np.abs(ev - evg)) <= rel * aN
    thr = max(g["kwargs"].get("eps", 0) or 0, 0) * aN
    if thr > 0:
        assert np.all(r.resNorms <= thr * (1 + 1e-6)) and np.all(np.array(g["resNorms"]) <= thr * (1 + 1e-6))
    its, itsg = r.stats["numOuterIterations"], g["stats"]["numOuterIterations"]
    tol = LOOSE.get(name, 0.0)
    assert abs(its - itsg) <= tol * itsg, (its, itsg)
    if name in LOOSE_MATVECS:
        assert abs(r.stats["numMatvecs"] - g["stats"]["numMatvecs"]) <= LOOSE_MATVECS[name] * g["stats"]["numMatvecs"]
    if tol == 0.0:
        assert r.stats["numMatvecs"] == g["stats"]["numMatvecs"]
        assert r.stats["numRestarts"] == g["stats"]["numRestarts"]
    if tol == 0.0 and its == itsg and r.stats["numMatvecs"] == g["stats"]["numMatvecs"]:
        # same convergence history: the residual norms themselves must be the reference's (north
        # star: eigenvalues AND residual norms within 1e-10 |A| in double, 1e-4 |A| in float)
        assert np.max(np.abs(np.array(r.resNorms, dtype=np.float64) - np.array(g["resNorms"]))) <= rel * aN


@pytest.mark.parametrize("name", ["lap2d_closest_abs", "ref_closest_geq", "harm_two_shifts", "jdqmr_blk4", "jdqmr_closest_abs", "lap2d_dynamic_few_soft"])
def test_invariant_subspace_against_the_references_eigenvectors(built, name):
    """CPU-checker leg of tests/test_solver_gpu.py::test_hip_invariant_subspace_against_the_references_eigenvectors (a sample of
    the 17 fixtures; the GPU suite runs all of them): the subspace returned for a fixture whose history is not the reference's
    count for count is the reference's own (tests/golden/reference_evecs.npz) to 2 eps |A| / gap."""
    from test_solver_gpu import subspace_check
    r, g = _run(name, "hostcheck")
    assert r.ret == 0 and r.initSize == g["initSize"]
    subspace_check(name, r, g)


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("dims,tgt,sh", [((20, 21), "smallest", 0.0), ((20, 21), "largest", 8.0), ((9, 10, 11), "largest", 12.0), ((20, 21), "largest_abs", 8.0)])
def test_refined_extraction_with_an_extremal_target_against_live_reference(built, dims, tgt, sh):
    """Refined extraction with an EXTREMAL target (round 6; returned -44 before): the reference lets it through
    (primme_c.c:512-520) and runs the refined procedure around targetShifts[0] — the pairs it converges to are the ones whose
    refined residual the target ordering prefers, not the ones the name of the target suggests (smallest + shift 0 ends at the
    top of the spectrum).  Whatever one thinks of that, a drop-in does the same: the same set of eigenvalues as the live
    reference, iteration counts within 8 %.  (Harmonic extraction with an extremal target stays refused: the reference's own
    solve_H_Harm has no case for it, solve_projection.c:469-482 `default: assert(0)`.)"""
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    aN = 8.0 if len(dims) == 2 else 12.0
    kw = dict(numEvals=4, target=tgt, targetShifts=[sh], eps=1e-9, aNorm=aN, projection="refined", v0=problems.start_vector(n))
    a = eigsh(op, backend="reference", **kw)
    b = eigsh(op, backend="hostcheck", **kw)
    assert a.ret == b.ret == 0 and a.initSize == b.initSize == 4
    assert np.max(np.abs(np.sort(a.evals) - np.sort(b.evals))) <= 1e-10 * aN
    assert np.all(b.resNorms <= 1e-9 * aN * (1 + 1e-6))
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= max(3, 0.08 * a.stats["numOuterIterations"])
    h = eigsh(op, backend="hostcheck", **dict(kw, projection="harmonic"))
    assert h.ret == -44


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_wide_basis_against_live_reference(built):
    """maxBasisSize beyond 255 (round 6: up to 1 023 on real panels; the reference has no limit, primme_c.c:470-487): basis 260,
    restart 60 on a 50 x 60 Laplacian, two restarts through the wide-basis update — the reference's eigenvalues, its restart
    count, its iteration count to 3 % (the histories of two bases this wide separate at rounding level)."""
    dims = (50, 60)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=8, eps=1e-11, aNorm=8.0, maxBasisSize=260, minRestartSize=60, v0=problems.start_vector(n), maxBlockSize=1)
    a = eigsh(op, backend="reference", **kw)
    b = eigsh(op, backend="hostcheck", **kw)
    assert a.ret == b.ret == 0 and b.initSize == 8 and b.params["maxBasisSize"] == 260
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * 8.0
    assert np.max(np.abs(b.evals - problems.laplacian_eigenvalues(dims, 8))) <= 1e-10 * 8.0
    assert b.stats["numRestarts"] == a.stats["numRestarts"] >= 2
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.03 * a.stats["numOuterIterations"]


@pytest.mark.parametrize("name", ["gen_gdk", "gen_gdk_blk2", "gen_largest_soft", "gen_olsen_jacobi", "gen_gd", "gen_lobpcg", "gen_blk4_3d", "gen_noanorm",
                                  "gen_lund_gdk", "gen_lund_jdqmr", "gen_lund_blk2"])
def test_generalized_problem_against_reference_fixture(built, name):
    """Round 6 widening (VERDICT r05 Missing #2): A x = lambda B x with massMatrixMatvec (primme_eigs.h:182-185) — the tracked
    V'BV block path with B applied through the callback.  Against the reference's own generalised solves (eigenvalues,
    residual norms, outer-iteration and restart counts — exact for the Generalized-Davidson fixtures), against scipy's dense
    truth, and the returned vectors B-orthonormal with the true residual A x - lambda B x."""
    from generalized_cases import check
    check(name, "hostcheck")


@pytest.mark.parametrize("name", ["gen_jdqmr", "gen_jdqmr_jacobi", "gen_jdqmr_etol_3d", "gen_jdqmr_largest", "gen_jdqmr_blk3", "gen_jd_olsen", "gen_jdqmr_soft"])
def test_generalized_jdqmr_against_reference_fixture(built, name):
    """The JDQMR inner solver on A x = lambda B x (round 6, csrc/eigs_jd.c): (A - sigma B) d with B through the callback, left
    projectors I - (B Q) Q' and I - (B x) x', right projectors on B evecs / K^-1 B evecs / K^-1 B x, the B-norm of the correction and
    |B x|^2 in the adaptive stopping tests (reference correction.c:862-997, inner_solve.c:283-303, :416-422, :838-890).  With locking
    and block size 1 the reference's outer iterations, restarts, preconditioner applications and residual norms are reproduced
    exactly; see generalized_cases.py for the block and the no-locking fixtures."""
    from generalized_cases import check
    check(name, "hostcheck")


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("method,mass", [("JDQMR", False), ("JDQMR_ETol", False), ("JDQMR/full_LTolerance", False), ("JDQMR", True)])
def test_inner_iteration_events_follow_the_live_reference(built, method, mass):
    """monitorFun receives primme_event_inner_iteration for every QMR step (reference inner_solve.c:550-558 with the adaptive
    tests, :581-588 otherwise): the same number of reports as dprimme, each with its outer-iteration count, inner step, eigenvalue
    estimate, residual estimate and QMR residual (1e-7 relative, 1e-4 with a mass matrix: the steps themselves are the reference's)."""
    import ctypes as C
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    op, bop = Operator(n, csr=(rp, ci, va)), Operator(n, csr=(brp, bci, bva))

    def run(be):
        log = []

        def mon(bev, bs, bf, iblock, blockSize, bnorms, numConv, lev, numLocked, lflags, lnorms, inner, lsres, msg, time, event, pp, ierr):
            if event[0] == 1:        # primme_event_inner_iteration
                assert bs[0] == 1 and blockSize[0] == 1 and iblock[0] == 0 and bf[0] == 0
                val = [C.cast(q, C.POINTER(C.c_double))[0] for q in (bev, bnorms, lsres)]
                log.append((pp[0].stats.numOuterIterations, inner[0], numConv[0], numLocked[0], *val))
            ierr[0] = 0
        def tw(p):       # the non-adaptive branch of the inner stopping tests (inner_solve.c:563-588): at most 12 steps to a full tolerance
            if "/" in method:
                p.correctionParams.convTest = 0          # primme_full_LTolerance
                p.correctionParams.maxInnerIterations = 12
        r = eigsh(op, backend=be, mass=bop if mass else None, v0=problems.start_vector(n), numEvals=4, eps=1e-9, aNorm=8.0, method=method.split("/")[0],
                  precond="jacobi", locking=1, monitor=mon, tweak=tw)
        assert r.ret == 0
        return r, log
    (a, la), (b, lb) = run("reference"), run("hostcheck")
    assert len(la) == len(lb) > 50 and [e[:4] for e in la] == [e[:4] for e in lb]
    # (with B the residual estimate of a step carries the rounding of two more operator applications)
    assert max(abs(x - y) / max(abs(x), 1e-300) for ea, eb in zip(la, lb) for x, y in zip(ea[4:], eb[4:])) <= (1e-4 if mass else 1e-7)
    assert a.stats["numOuterIterations"] == b.stats["numOuterIterations"] and a.stats["numPreconds"] == b.stats["numPreconds"]


def test_generalized_block_jdqmr_with_the_references_own_indexing(built, monkeypatch):
    """PRIMME_AMD_JDQMR_REF_INDEXING=1 on a generalised problem: besides its indexing of the block recurrences the reference leaves
    the B x panel unpermuted when a column leaves the block (inner_solve.c:352-357 permutes x and the right projector only);
    restated under the knob, the block fixture's counts become exact."""
    monkeypatch.setenv("PRIMME_AMD_JDQMR_REF_INDEXING", "1")
    from generalized_cases import check
    check("gen_jdqmr_blk3", "hostcheck")


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_generalized_jdqmr_without_locking_the_references_way(built, monkeypatch):
    """PRIMME_AMD_JDQMR_REF_SOFT_LOCKING=1: without locking the reference allocates no B evecs and its projectors read evecs in
    their place (main_iter.c:334-336, :659-661; for block size 1, correction.c:911-917 then overwrites x with B x).  Under the
    knob the inner history is the reference's step for step — eigenvalue estimate, residual estimate and QMR residual of every
    reported inner step to 1e-6 through the first thirteen outer iterations (after that the reference's inner solve stagnates for
    hundreds of steps and the exit is decided at rounding level); the default keeps B evecs and needs well under half the work."""
    import ctypes as C
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    op, bop = Operator(n, csr=(rp, ci, va)), Operator(n, csr=(brp, bci, bva))
    logs = {}

    def run(be):
        log = []

        def mon(bev, bs, bf, iblock, blockSize, bnorms, numConv, lev, numLocked, lflags, lnorms, inner, lsres, msg, time, event, pp, ierr):
            if event[0] == 1:        # primme_event_inner_iteration
                val = [C.cast(q, C.POINTER(C.c_double))[0] for q in (bev, bnorms, lsres)]
                log.append((pp[0].stats.numOuterIterations, inner[0], *val))
            ierr[0] = 0
        r = eigsh(op, backend=be, mass=bop, v0=problems.start_vector(n), numEvals=5, eps=1e-9, aNorm=8.0, method="JDQMR", precond="jacobi",
                  locking=0, monitor=mon)
        assert r.ret == 0
        return r, log
    plain, _ = run("hostcheck")
    monkeypatch.setenv("PRIMME_AMD_JDQMR_REF_SOFT_LOCKING", "1")
    ref, a = run("reference")
    mine, b = run("hostcheck")
    na = sum(1 for e in a if e[0] <= 13)
    assert na > 200 and [e[:2] for e in a[:na]] == [e[:2] for e in b[:na]]
    assert max(abs(x - y) / abs(x) for ea, eb in zip(a[:na], b[:na]) for x, y in zip(ea[2:], eb[2:])) <= 1e-6
    assert abs(mine.stats["numOuterIterations"] - ref.stats["numOuterIterations"]) <= 2
    assert np.max(np.abs(mine.evals - ref.evals)) <= 1e-10 * 8.0 and np.max(np.abs(plain.evals - ref.evals)) <= 1e-10 * 8.0
    assert plain.stats["numMatvecs"] < 0.5 * ref.stats["numMatvecs"]


SINGLE_GEN = [dict(method="GD_plusK"), dict(method="JDQMR", locking=1, precond="jacobi"), dict(method="GD_plusK", maxBlockSize=2)]


def _single_generalized(backend, kw):
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    return eigsh(Operator(n, csr=(rp, ci, va.astype(np.float32))), backend=backend, mass=Operator(n, csr=(brp, bci, bva.astype(np.float32))),
                 v0=problems.start_vector(n).astype(np.float32), dtype=np.float32, numEvals=4, eps=1e-4, aNorm=8.0, **kw)


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("kw", SINGLE_GEN)
def test_generalized_single_precision_against_live_reference(built, kw):
    """sprimme with a mass matrix (single-precision panels take the tracked-Gram block path anyway): the live reference's outer
    iterations and restarts exactly, its eigenvalues to single-precision rounding."""
    a, b = _single_generalized("reference", kw), _single_generalized("hostcheck", kw)
    assert a.ret == b.ret == 0 and b.initSize == 4
    assert (a.stats["numOuterIterations"], a.stats["numRestarts"]) == (b.stats["numOuterIterations"], b.stats["numRestarts"])
    assert np.max(np.abs(a.evals - b.evals)) <= 2e-6 * 8.0


def test_generalized_problem_corners(built, monkeypatch):
    """The corners of the mass-matrix path: the dynamic method switches between GD+k and JDQMR with B like on a standard problem
    (PRIMME_AMD_MASS_NO_DYNAMIC=1 keeps it in GD+k and says so), exact Olsen and skew projectors work on K^-1 B x, and a
    non-Rayleigh-Ritz projection is an input error like in the reference (-39, primme_c.c:518-520)."""
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    brp, bci, bva = problems.mass_matrix_csr(n)
    op, bop = Operator(n, csr=(rp, ci, va)), Operator(n, csr=(brp, bci, bva))
    kw = dict(numEvals=2, eps=1e-8, aNorm=8.0, v0=problems.start_vector(n))
    import scipy.linalg as sl, scipy.sparse as sp
    w = sl.eigh(sp.csr_matrix((va, ci, rp), shape=(n, n)).toarray(), sp.csr_matrix((bva, bci, brp), shape=(n, n)).toarray(), eigvals_only=True)[:2]
    for method, extra in (("DYNAMIC", {}), ("JDQR", dict(precond="jacobi")), ("GD_Olsen_plusK", dict(precond="jacobi")), ("JDQMR", dict(locking=0))):
        d = eigsh(op, backend="hostcheck", mass=bop, method=method, **kw, **extra)
        assert d.ret == 0 and np.max(np.abs(d.evals - w)) <= 1e-9 * 8.0, method

    def skew_x(p):       # exact Olsen: (I - K^-1 B x x' / x'K^-1 B x) K^-1 r  (correction.c:700-777)
        p.correctionParams.projectors.RightX = 1
        p.correctionParams.projectors.SkewX = 1
    d = eigsh(op, backend="hostcheck", mass=bop, method="GD_plusK", precond="jacobi", tweak=skew_x, **kw)
    assert d.ret == 0 and np.max(np.abs(d.evals - w)) <= 1e-9 * 8.0
    monkeypatch.setenv("PRIMME_AMD_MASS_NO_DYNAMIC", "1")
    d = eigsh(op, backend="hostcheck", mass=bop, method="DYNAMIC", **kw)
    assert d.ret == 0 and d.params["dynamicMethodSwitch"] == -2 and np.max(np.abs(d.evals - w)) <= 1e-9 * 8.0
    assert eigsh(op, backend="hostcheck", mass=bop, projection="refined", target="closest_abs", targetShifts=[1.0], **kw).ret == -39


def test_block_jdqmr_with_the_references_own_indexing(built, monkeypatch):
    """PRIMME_AMD_JDQMR_REF_INDEXING=1 (csrc/eigs_jd.c): the block QMR recurrences indexed the way the reference indexes them —
    sigma_prev, Theta and rho written by block position, read by original column, x permuted once more per projector it doubles
    as (reference inner_solve.c:317, :329-337, :352-357, :373-377, :600-603, :616-620), in the reference's operation order.
    With it the block fixture reproduces dprimme's outer-iteration, matvec and restart counts EXACTLY and then its residual
    norms; without it (the default: every recurrence stays with its own column) the same problem takes a different, equally
    valid history — the reason the fixture tests carry a 15 % / 30 % tolerance for block JDQMR."""
    r0, g = _run("jdqmr_blk4", "hostcheck")
    monkeypatch.setenv("PRIMME_AMD_JDQMR_REF_INDEXING", "1")
    r, g = _run("jdqmr_blk4", "hostcheck")
    got = (r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"])
    want = (g["stats"]["numOuterIterations"], g["stats"]["numMatvecs"], g["stats"]["numRestarts"])
    assert r.ret == 0 and got == want, (got, want)
    assert np.max(np.abs(np.array(r.evals) - np.array(g["evals"]))) <= 1e-10 * 8.0
    assert np.max(np.abs(np.array(r.resNorms) - np.array(g["resNorms"]))) <= 1e-10 * 8.0
    assert (r0.stats["numOuterIterations"], r0.stats["numMatvecs"]) != want[:2]      # the default is the other iteration
    # the preconditioned block-8 fixture comes closer with it, not exact (near-degenerate 3-D spectrum: the histories part at rounding level)
    r8, g8 = _run("jdqmr_etol_blk8_jacobi", "hostcheck")
    assert abs(r8.stats["numMatvecs"] - g8["stats"]["numMatvecs"]) <= 0.02 * g8["stats"]["numMatvecs"]
    assert abs(r8.stats["numOuterIterations"] - g8["stats"]["numOuterIterations"]) <= 3


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("dims,method,b,pc", [((20, 21, 22), "JDQMR", 2, "jacobi"), ((20, 21, 22), "JDQMR", 4, None), ((20, 21, 22), "JDQMR_ETol", 4, "jacobi"),
                                              ((30, 31), "JDQMR", 8, None), ((30, 31), "JDQMR_ETol", 8, "jacobi"), ((30, 31), "JDQMR_ETol", 2, None)])
def test_block_jdqmr_reference_indexing_against_live_reference(built, monkeypatch, dims, method, b, pc):
    """The same knob against the LIVE reference at block sizes 2, 4 and 8, with and without the Jacobi preconditioner: identical
    outer-iteration / matvec / restart counts (20 of 24 configurations of the sweep in profiles/r06_jdqmr_reference_indexing.txt
    are exact; the four that are not are block size 8 on the near-degenerate 3-D spectrum, within 3 %)."""
    monkeypatch.setenv("PRIMME_AMD_JDQMR_REF_INDEXING", "1")
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=10 if len(dims) == 3 else 6, method=method, eps=1e-9, aNorm=12.0 if len(dims) == 3 else 8.0, maxBlockSize=b, v0=problems.start_vector(n))
    if pc:
        kw["precond"] = pc
    a = eigsh(op, backend="reference", **kw)
    c = eigsh(op, backend="hostcheck", **kw)
    f = lambda r: (r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"])
    assert a.ret == c.ret == 0 and f(a) == f(c), (f(a), f(c))
    assert np.max(np.abs(a.evals - c.evals)) <= 1e-10 * kw["aNorm"]
    assert np.max(np.abs(a.resNorms - c.resNorms)) <= 1e-10 * kw["aNorm"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_block_and_guesses_against_live_reference(built, seed):
    """Block size 2 (implicit_I forced) and more initial guesses than fit, from random starts."""
    dims = (20, 21)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    rng = np.random.default_rng(5 + seed)
    vr = rng.standard_normal((n, 9))
    for kw in (dict(numEvals=5, eps=1e-9, aNorm=8.0, v0=vr[:, :2], maxBlockSize=2, orth=F.primme_orth_implicit_I),
               dict(numEvals=4, eps=1e-9, aNorm=8.0, v0=vr, maxBasisSize=12, minRestartSize=5)):
        a = eigsh(op, backend="reference", **kw)
        b = eigsh(op, backend="hostcheck", **kw)
        assert a.ret == b.ret == 0
        assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * 8.0
        assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.05 * a.stats["numOuterIterations"]


def test_analytic_spectrum_and_true_residuals(built):
    """check_solution semantics of the reference's test driver (tests/COMMON/ioandtest.c:96-145):
    orthonormal vectors, Rayleigh quotients, recomputed residual norms agree with the reported."""
    dims = (17, 19, 13)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    r = eigsh(op, backend="hostcheck", numEvals=8, eps=1e-9, aNorm=12.0, v0=problems.start_vector(n))
    assert r.ret == 0
    exact = problems.laplacian_eigenvalues(dims, 8)
    assert np.max(np.abs(r.evals - exact)) <= 1e-10 * 12.0
    V = r.evecs
    assert np.max(np.abs(V.T @ V - np.eye(8))) < 1e-7
    AV = problems.csr_matvec_numpy(rp, ci, va, V)
    for i in range(8):
        rq = V[:, i] @ AV[:, i]
        true_rn = np.linalg.norm(AV[:, i] - r.evals[i] * V[:, i])
        assert abs(rq - r.evals[i]) <= max(r.resNorms[i], 12.0 * 1e-14)
        assert abs(true_rn - r.resNorms[i]) <= max(2 * true_rn, 10 * 12.0 * 2.2e-16) and true_rn <= 1e-9 * 12.0 * 1.01


def test_edge_cases(built):
    # n smaller than the default basis; all eigenpairs of a tiny matrix; numEvals = 0; constraints
    rp, ci, va, n = problems.laplacian_csr((7,))
    op = Operator(n, csr=(rp, ci, va))
    r = eigsh(op, backend="hostcheck", numEvals=3, eps=1e-12, aNorm=4.0, v0=problems.start_vector(n))
    assert r.ret == 0 and np.max(np.abs(r.evals - problems.laplacian_eigenvalues((7,), 3))) < 1e-12
    r = eigsh(op, backend="hostcheck", numEvals=7, eps=1e-12, aNorm=4.0, v0=problems.start_vector(n))
    assert r.ret == 0 and np.max(np.abs(np.sort(r.evals) - problems.laplacian_eigenvalues((7,), 7))) < 1e-12
    r = eigsh(op, backend="hostcheck", numEvals=0)
    assert r.ret == 0 and r.initSize == 0
    # maxMatvecs reached: returns -3 (PRIMME_MAIN_ITER_FAILURE) with the best candidates
    rp, ci, va, n = problems.laplacian_csr((30, 31))
    op = Operator(n, csr=(rp, ci, va))
    r = eigsh(op, backend="hostcheck", numEvals=5, eps=1e-12, aNorm=8.0, v0=problems.start_vector(n), maxMatvecs=40)
    assert r.ret == -3 and r.stats["numMatvecs"] <= 41
    # configurations that are not on the device path fail loudly with -44 (a basis beyond 1 023 columns; 255 until round 6)
    rpw, ciw, vaw, nw = problems.laplacian_csr((40, 41))
    r = eigsh(Operator(nw, csr=(rpw, ciw, vaw)), backend="hostcheck", numEvals=2, maxBasisSize=1100, aNorm=8.0, v0=problems.start_vector(nw))
    assert r.ret == -44


def test_dynamic_method_leaves_a_recommendation(built):
    """PRIMME_DYNAMIC (reference main_iter.c:427-436, :1221-1228): converges like the fixed methods
    and reports -1 (GD+k), -2 (JDQMR_ETol) or -3 (close call) in dynamicMethodSwitch."""
    dims = (100,)
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    r = eigsh(op, backend="hostcheck", numEvals=10, eps=1e-9, aNorm=4.0, precond="jacobi", method="DYNAMIC",
              v0=problems.start_vector(n))
    assert r.ret == 0 and r.params["dynamicMethodSwitch"] in (-1, -2, -3)
    assert np.max(np.abs(r.evals - problems.laplacian_eigenvalues(dims, 10))) <= 1e-10 * 4.0
    assert np.all(r.resNorms <= 1e-9 * 4.0 * (1 + 1e-6))


@pytest.mark.parametrize("mode", ["no_wtr", "no_fused_restart", "no_spec_restart", "default"])
def test_launch_structure_block_size_one(built, mode, monkeypatch):
    """GD+k, block size 1, no preconditioner: per outer iteration ONE fused residual+overlaps
    pass, ONE Gram-Schmidt update (speculative, reused by the orthogonaliser), and for the
    projection either none (default): the column comes from W'r of the fused pass and only t'At is a
    (two-vector) inner product; or, with PRIMME_AMD_NO_WTR=1, ONE inner-product pass over V (the
    reference's formula).  Default: the iteration after a restart has the same structure and the check at
    the full basis is the restart pass itself (eigs_solver.h: speculative restart);
    PRIMME_AMD_NO_FUSED_RESTART=1: a norm-only pass before and three panel products after every restart.
    Iteration / restart counts are the same in all modes."""
    import ctypes as C
    if mode == "no_wtr":
        monkeypatch.setenv("PRIMME_AMD_NO_WTR", "1")
    if mode == "no_fused_restart":
        monkeypatch.setenv("PRIMME_AMD_NO_FUSED_RESTART", "1")
    if mode == "no_spec_restart":
        monkeypatch.setenv("PRIMME_AMD_NO_SPEC_RESTART", "1")
    wtr = mode != "no_wtr"
    lib = checkers.load_hostcheck()
    cnt = (C.c_long * 8)()
    lib.hipk_cpu_counts(cnt, 1)
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", numEvals=10, eps=1e-10, aNorm=8.0,
              v0=problems.start_vector(n))
    lib.hipk_cpu_counts(cnt, 1)
    its, rst = r.stats["numOuterIterations"], r.stats["numRestarts"]
    dots, project, ritz_upd, ritz_cgs, fused_tail, ritz_ov = cnt[0], cnt[1], cnt[2], cnt[3], cnt[5], cnt[6]
    assert r.ret == 0 and its == 490 and rst == 69 and r.stats["numMatvecs"] == 490
    # Round 5: the next iteration is enqueued before the host has seen the current one (eigs_conv.c: pa_prelaunch_next) and thrown
    # away when the host then decides otherwise — the candidate converged, a pair gets locked: a handful per solve (one per
    # wanted pair here).  Those launches are counted by the checker but are no iterations: `ahead` is their allowance.
    ahead = 12
    fused_tail -= min(ahead, max(0, fused_tail - its)); ritz_cgs -= min(ahead, max(0, ritz_cgs - (its - rst if mode == "default" else its)))
    project -= min(ahead, max(0, project - its))
    if mode == "default":
        # every iteration but the ones around a locked pair: one-launch tail, fused pass; the check at the full
        # basis IS the restart pass (speculative, out of place; with locked pairs plus one panel product for
        # Q'r and W(:,k-1)'Q): a separate restart pass remains only where a pair gets locked
        assert its - 15 <= fused_tail <= its
        assert rst - 12 <= ritz_ov <= rst and ritz_upd <= 30
        assert its - rst - 15 <= ritz_cgs <= its - rst + 30
        assert dots <= rst + 45
    elif mode == "no_spec_restart":
        # the check at the full basis is a fused residual pass on the old basis, the restart pass follows
        assert its - 15 <= fused_tail <= its and ritz_ov == 0
        assert its <= ritz_cgs <= its + 25 and dots <= 45
    else:
        # with the library's own operator the tail of the iteration (normalise, A t, t'At) is ONE launch
        assert (fused_tail >= its - rst - 15 and fused_tail <= its) if wtr else fused_tail == 0
        assert ritz_cgs >= its - rst - 15 and ritz_cgs <= its
        if not wtr:
            assert dots <= its + rst + 25         # projection pass each iteration + CGS dots after restarts
        else:
            assert dots <= 3 * rst + 40           # panel inner products only around restarts / second passes
    assert project <= its + 25               # one update per new vector (+ rare second passes)


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("dims,nev", [((20, 21), 10), ((20, 21), 4), ((9, 10, 11), 7)])
def test_application_matvec_callback_keeps_the_restart_paths(built, dims, nev):
    """with an application's matvec callback the one-launch tail is not available, the fused / speculative restart is:
    counts are the reference's, the restart pass is the speculative one, no fused-tail launches"""
    import ctypes as C
    lib = checkers.load_hostcheck()
    rp, ci, va, n = problems.laplacian_csr(dims)

    def matvec(x, ldx, y, ldy, bs, pp, ierr):
        nb, lx, ly = bs[0], ldx[0], ldy[0]
        X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(nb, lx))
        Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(nb, ly))
        Y[:, :n] = problems.csr_matvec_numpy(rp, ci, va, X[:, :n].T).T
        ierr[0] = 0
    cb = F.BLOCK_OP(matvec)
    cnt = (C.c_long * 8)()
    kw = dict(numEvals=nev, eps=1e-10, aNorm=4.0 * len(dims), v0=problems.start_vector(n), method="GD_plusK")
    lib.hipk_cpu_counts(cnt, 1)
    h = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", user_matvec=cb, **kw)
    lib.hipk_cpu_counts(cnt, 1)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", **kw)
    assert h.ret == 0 and r.ret == 0
    for key in ("numOuterIterations", "numMatvecs", "numRestarts"):
        assert h.stats[key] == r.stats[key], key
    assert cnt[5] == 0 and cnt[6] >= r.stats["numRestarts"] - 12 and cnt[0] <= r.stats["numRestarts"] + 45


# ---- the reference's own driver regression cases on LUNDA.mtx -----------------------------
import reference_driver_cases as RD


def _run_driver_case(name, backend):
    rp, ci, va, n = RD.lunda()
    op = Operator(n, csr=(rp, ci, va))
    case = RD.CASES[name]
    r = eigsh(op, backend=backend, **case["kw"])
    X = RD.read_sol(case["sol"], n)
    bad = RD.check_solution(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                            r.evals, np.asarray(r.evecs, dtype=np.float64), r.resNorms, r.params["aNorm"],
                            case["kw"]["eps"], X)
    return r, bad


@pytest.mark.parametrize("name", sorted(RD.CASES))
def test_reference_driver_case(built, name):
    """tests/tests/test_00N through the product host solver; accepted by the reference driver's
    check_solution against the reference's stored eigenvectors sol_00N_double."""
    r, bad = _run_driver_case(name, "hostcheck")
    assert r.ret == 0 and r.initSize == RD.CASES[name]["kw"]["numEvals"]
    assert not bad, bad
    # the spectrum itself: dense truth
    rp, ci, va, n = RD.lunda()
    A = np.zeros((n, n))
    A[np.repeat(np.arange(n), np.diff(rp)), ci] = va
    w = np.linalg.eigvalsh(A)
    k = len(r.evals)
    want = np.sort(w)[::-1][:k] if RD.CASES[name]["kw"]["target"] == "largest" else w[np.argsort(np.abs(w))][:k]
    assert np.max(np.abs(np.sort(r.evals) - np.sort(want))) <= 1e-10 * r.params["aNorm"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", sorted(RD.CASES))
def test_reference_driver_case_pins_the_checker(built, name):
    """The same acceptance test applied to the live reference build: pins check_solution's
    restatement and the data files (if this failed, the test above would prove nothing)."""
    r, bad = _run_driver_case(name, "reference")
    assert r.ret == 0 and not bad, bad
    h, _ = _run_driver_case(name, "hostcheck")
    assert np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) <= 1e-10 * r.params["aNorm"]
    if name == "test_001":
        # unrestarted: the product host logic follows the reference step for step
        for k in ("numOuterIterations", "numMatvecs", "numRestarts"):
            assert h.stats[k] == r.stats[k], k
    else:
        # eps = 1e-12 on a matrix with cond 3e6 converges at the rounding-noise floor (residuals
        # 1e-12 |A|), where the summation order of the inner products decides the last steps
        assert abs(h.stats["numOuterIterations"] - r.stats["numOuterIterations"]) <= 0.15 * r.stats["numOuterIterations"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_orthogonality_constraints_against_live_reference(built):
    """numOrthoConst > 0 (reference primme_eigs.h: the first numOrthoConst columns of evecs are
    constraints): same path, same counts, results orthogonal to the constraints."""
    import ctypes as C
    dims = (20, 21)
    rp, ci, va, n = problems.laplacian_csr(dims)
    # constraints = the two lowest eigenvectors (analytic): the solver must return pairs 3..6
    def evec(i, j):
        x = np.sin(np.pi * i * np.arange(1, dims[0] + 1) / (dims[0] + 1))
        y = np.sin(np.pi * j * np.arange(1, dims[1] + 1) / (dims[1] + 1))
        v = np.outer(y, x).ravel()
        return v / np.linalg.norm(v)
    Q = np.stack([evec(1, 1), evec(1, 2)], axis=1)
    exact = problems.laplacian_eigenvalues(dims, 6)
    out = {}
    for be in ("hostcheck", "reference"):
        lib = checkers.load_hostcheck() if be == "hostcheck" else checkers.load_reference()
        p = F.PrimmeParams()
        lib.primme_initialize(C.byref(p))
        p.n, p.numEvals, p.eps, p.aNorm, p.numOrthoConst, p.printLevel, p.outputFile = n, 4, 1e-10, 8.0, 2, 0, None
        keep = []
        if be == "reference":
            def mv(x, ldx, y, ldy, bs, pp, ierr):
                nb = bs[0]
                X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(nb, ldx[0]))
                Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(nb, ldy[0]))
                Y[:, :n] = problems.csr_matvec_numpy(rp, ci, va, X[:, :n].T).T
                ierr[0] = 0
            cb = F.BLOCK_OP(mv); keep.append(cb)
            p.matrixMatvec = C.cast(cb, C.c_void_p)
            solver = lib.dprimme
        else:
            ctx = C.c_void_p(); lib.hipk_ctx_create(C.byref(ctx), None)
            A = C.c_void_p()
            lib.hipk_csr_create(ctx, F.HIPK_F64, n, n, 0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                va.ctypes.data_as(C.c_void_p), C.byref(A))
            oph = C.c_void_p(); lib.primme_amd_operator_create(C.byref(oph), A, None)
            p.matrix = oph
            p.matrixMatvec = C.cast(lib.primme_amd_matvec, C.c_void_p)
            solver = lib.hip_dprimme
        lib.primme_set_method(F.METHODS["GD_plusK"], C.byref(p))
        ev, rn, vecs = np.zeros(4), np.zeros(4), np.zeros((6, n))
        vecs[:2] = Q.T
        ret = solver(ev.ctypes.data_as(C.c_void_p), vecs.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p))
        assert ret == 0 and p.initSize == 4
        assert np.max(np.abs(ev - exact[2:6])) <= 1e-10 * 8.0
        assert np.linalg.norm(vecs[2:6] @ Q) <= 1e-9 and np.allclose(vecs[:2], Q.T, atol=1e-13)
        out[be] = (ev.copy(), p.stats.numOuterIterations, p.stats.numMatvecs, p.stats.numRestarts)
    assert np.max(np.abs(out["hostcheck"][0] - out["reference"][0])) <= 1e-10 * 8.0
    assert out["hostcheck"][1:] == out["reference"][1:]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_exact_olsen_against_live_reference(built):
    """JD_Olsen_plusK (RightX + SkewX with maxInnerIterations = 0): the exact Olsen correction
    K^-1 r - (x'K^-1 r / x'K^-1 x) K^-1 x (reference correction.c:718-777) with a non-trivial
    diagonal preconditioner (LUNDA.mtx, K = diag(A) - 3e8)."""
    rp, ci, va, n = RD.lunda()
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=4, eps=1e-10, target="largest", method="JD_Olsen_plusK", precond=("jacobi", 3e8))
    a = eigsh(op, backend="reference", **kw)
    b = eigsh(op, backend="hostcheck", **kw)
    assert a.ret == 0 and b.ret == 0
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * a.params["aNorm"]
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.05 * a.stats["numOuterIterations"] + 2
    assert a.stats["numPreconds"] > a.stats["numMatvecs"]      # two preconditioner applications per step
    assert abs(a.stats["numPreconds"] - b.stats["numPreconds"]) <= 0.05 * a.stats["numPreconds"] + 4


def _lunda_dense():
    rp, ci, va, n = RD.lunda()
    A = np.zeros((n, n))
    A[np.repeat(np.arange(n), np.diff(rp)), ci] = va
    return rp, ci, va, n, np.linalg.eigvalsh(A)


@pytest.mark.parametrize("kw", [dict(numEvals=6, method="JDQR"), dict(numEvals=6, method="JDQR", locking=0),
                                dict(numEvals=6, method="JDQR", maxBlockSize=2)])
def test_skew_projectors(built, kw):
    """K^-1-weighted right projectors of the correction equation (reference inner_solve.c:714-808,
    correction.c:942-983, restart.c:1471-1531): JDQR with a diagonal preconditioner on LUNDA.mtx."""
    rp, ci, va, n, w = _lunda_dense()
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", eps=1e-10, target="largest", precond=("jacobi", 3e8), **kw)
    k = kw["numEvals"]
    assert r.ret == 0 and r.initSize == k
    assert np.max(np.abs(np.sort(r.evals) - np.sort(w)[-k:])) <= 1e-10 * r.params["aNorm"]
    assert np.all(r.resNorms <= 1e-10 * r.params["aNorm"] * (1 + 1e-6))
    assert r.stats["numPreconds"] > 0


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_skew_projector_single_pair_against_live_reference(built):
    """One wanted pair keeps M = q'K^-1 q scalar; for more stored vectors the reference as built here
    crashes (it hands its factorisation a leading dimension of 0: main_iter.c:417, :1089 ->
    factorize.c:228-236), so only this case can be compared count for count."""
    rp, ci, va, n = RD.lunda()
    op = Operator(n, csr=(rp, ci, va))
    kw = dict(numEvals=1, eps=1e-10, target="largest", method="JDQR", precond=("jacobi", 3e8))
    a = eigsh(op, backend="reference", **kw)
    b = eigsh(op, backend="hostcheck", **kw)
    assert a.ret == 0 and b.ret == 0 and abs(a.evals[0] - b.evals[0]) <= 1e-10 * a.params["aNorm"]
    for key in ("numOuterIterations", "numMatvecs", "numPreconds", "numRestarts"):
        assert a.stats[key] == b.stats[key], key


def test_returns_instead_of_spinning_when_the_space_is_exhausted(built):
    """n = 88 with 3 constraints, block size 20, single precision: after two outer iterations the basis
    spans everything that is left, nothing more can be added and the pairs do not meet the tolerance.
    The reference restarts the same basis forever here (and with maxBasisSize above the space it never
    leaves the inner loop); this solver must come back with PRIMME_MAIN_ITER_FAILURE."""
    rp, ci, va, n = problems.laplacian_csr((11, 8))
    rng = np.random.default_rng(5)
    Q = np.linalg.qr(rng.standard_normal((n, 3)))[0]
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=np.float32, method="GD_plusK", numEvals=20, eps=1e-4,
              iseed=(3200, 3490, 3396, 3989), target="largest", maxBlockSize=20, constraints=Q, maxMatvecs=15000)
    assert r.ret in (0, -3) and r.stats["numMatvecs"] <= 15000
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=np.float32, method="JD_Olsen_plusK", numEvals=30, eps=1e-4,
              iseed=(938, 3511, 1144, 2009), target="largest", maxBlockSize=6, maxBasisSize=150, precond=("jacobi", 0.2033),
              constraints=Q, maxMatvecs=15000)
    assert r.ret in (0, -3) and r.stats["numMatvecs"] <= 15000


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("projection", ["harmonic", "refined"])
@pytest.mark.parametrize("dtype,block,eps", [(np.float64, 3, 1e-9), (np.float32, 1, 1e-4), (np.float32, 3, 1e-4)])
def test_interior_extractions_with_explicit_I_against_live_reference(built, projection, dtype, block, eps, monkeypatch):
    """Harmonic / refined extraction when the basis is kept with a tracked Gram matrix (blocks, single
    precision: orth = explicit_I).  The reference tracks Q'Q as well (solve_projection.c:431-520,
    :542-560); here Q stays orthonormal by Gram-Schmidt with reorthogonalisation and only V'V enters
    the coefficient vectors.  Same eigenpairs, similar iteration counts."""
    rp, ci, va, n = problems.laplacian_csr((20, 21))
    A = np.zeros((n, n)); A[np.repeat(np.arange(n), np.diff(rp)), ci] = va
    w = np.linalg.eigvalsh(A)
    kw = dict(dtype=dtype, numEvals=4, target="closest_abs", targetShifts=[3.3], projection=projection, maxBlockSize=block,
              eps=eps, method="GD_Olsen_plusK", iseed=(1, 2, 3, 4), maxMatvecs=30000)
    ref = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", **kw)
    got = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", **kw)
    assert ref.ret == 0 and got.ret == 0 and got.initSize == 4
    assert got.params["orth"] == ref.params["orth"] == 2          # primme_orth_explicit_I
    aN = got.params["aNorm"]
    X = got.evecs.astype(np.float64)
    res = np.linalg.norm(A @ X - X * got.evals.astype(np.float64), axis=0)
    assert np.all(res <= 1.5 * eps * aN + 50 * np.finfo(dtype).eps * aN)
    tol = 1e-9 if dtype == np.float64 else 2e-3
    assert all(np.min(np.abs(w - ev)) <= tol * aN for ev in got.evals)
    assert np.max(np.abs(np.sort(got.evals) - np.sort(ref.evals))) <= (1e-8 if dtype == np.float64 else 5e-3) * aN
    # (refined: the singular vectors of R come from a different SVD algorithm than the reference's xGESVD; on this
    # clustered interior spectrum the histories separate early and the totals differ like two start vectors would)
    loose = 0.35 if projection == "refined" else 0.25
    assert abs(got.stats["numOuterIterations"] - ref.stats["numOuterIterations"]) <= loose * ref.stats["numOuterIterations"]


@pytest.mark.parametrize("dtype,eps", [(np.float64, 1e-9), (np.float32, 1e-4)])
def test_more_constraints_than_basis_columns_with_explicit_I(built, dtype, eps):
    """numOrthoConst > maxBasisSize with orth = explicit_I: init_basis orthonormalises the constraint
    block in one CholQR sweep, so the coefficient / reduction buffers must be sized for
    max(maxBasisSize, numOrthoConst), not for maxBasisSize (advisor finding, round 1: heap overflow)."""
    n, nc = 600, 20
    rp, ci, va, _ = problems.laplacian_csr((n,))
    j = np.arange(1, n + 1)
    Q = np.stack([np.sin(np.pi * k * j / (n + 1)) for k in range(1, nc + 1)], axis=1)
    Q /= np.linalg.norm(Q, axis=0)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=dtype, numEvals=3, eps=eps, aNorm=4.0,
              maxBasisSize=15, orth=F.primme_orth_explicit_I, constraints=Q, v0=problems.start_vector(n))
    assert r.ret == 0 and r.initSize == 3
    exact = 2 - 2 * np.cos(np.pi * np.arange(nc + 1, nc + 4) / (n + 1))
    assert np.max(np.abs(r.evals - exact)) <= (1e-10 if dtype == np.float64 else 1e-4) * 4.0


def test_single_precision_callbacks_get_float_operands(built, monkeypatch):
    """hip_sprimme hands globalSumReal / monitorFun operands of *_type, defaulted to float like the
    reference's sprimme (primme_c.c:170-183): an application whose callback reduces MPI_FLOAT keeps
    working (advisor finding, round 1: the callbacks used to receive doubles tagged as float)."""
    import ctypes as C
    monkeypatch.setenv("PRIMME_AMD_FORCE_COMM", "1")      # one rank, but every reduction goes through the callback
    rp, ci, va, n = problems.laplacian_csr((24, 25))
    seen = {"types": set(), "calls": 0, "mon": set()}

    def gsum(a):
        seen["calls"] += 1
        assert np.all(np.isfinite(a)) and np.all(np.abs(a) < 1e6)     # doubles read as floats would be garbage
        return a

    def mon(bev, bsz, bfl, ibl, blk, bno, ncv, lev, nlk, lfl, lno, inner, lsres, msg, tm, ev, pp, ierr):
        seen["mon"].add(pp[0].monitorFun_type)
        if bev and bsz[0] > 0:
            v = np.ctypeslib.as_array(C.cast(bev, C.POINTER(C.c_float)), shape=(bsz[0],))
            assert np.all(np.isfinite(v)) and np.all(np.abs(v) <= 8.0 * 1.01)
        ierr[0] = 0

    kw = dict(dtype=np.float32, numEvals=3, eps=1e-4, aNorm=8.0, v0=problems.start_vector(n))
    a = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", global_sum=gsum, monitor=mon, **kw)
    b = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", **kw)
    assert a.ret == 0 and b.ret == 0 and seen["calls"] > 10 and seen["mon"] == {F.primme_op_float}
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-4 * 8.0


def test_host_pointer_entry_points(built):
    """dprimme() with the reference's HOST-pointer contract (csrc/eigs_hostapi.c): host evecs, host
    callbacks that receive the CALLER's primme_params (matrix, ShiftsForPreconditioner ...).  Same
    solve as hip_dprimme with the ready-made operator: identical counts and eigenvalues (configs[0]:
    the ex_eigs_dseq problem)."""
    import ctypes as C
    lib = checkers.load_hostcheck()
    n, nev = 100, 10
    rp, ci, va, _ = problems.laplacian_csr((n,))
    seen = {"mv": 0, "pc": 0, "user_struct": True}
    p = F.PrimmeParams()
    lib.primme_initialize(C.byref(p))

    def mv(x, ldx, y, ldy, bs, pp, ierr):
        seen["mv"] += bs[0]
        seen["user_struct"] &= (C.addressof(pp[0]) == C.addressof(p))
        X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(bs[0], ldx[0]))
        Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(bs[0], ldy[0]))
        Y[:, :n] = problems.csr_matvec_numpy(rp, ci, va, X[:, :n].T).T
        ierr[0] = 0

    def pc(x, ldx, y, ldy, bs, pp, ierr):
        seen["pc"] += bs[0]
        X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(bs[0], ldx[0]))
        Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(bs[0], ldy[0]))
        sh = pp[0].ShiftsForPreconditioner
        for c in range(bs[0]):
            d = 2.0 - (sh[c] if sh else 0.0)
            if not abs(d) > 1e-14 * 4.0: d = np.copysign(1e-14 * 4.0, d)
            Y[c, :n] = X[c, :n] / d
        ierr[0] = 0
    cmv, cpc = F.BLOCK_OP(mv), F.BLOCK_OP(pc)
    p.n, p.numEvals, p.eps, p.aNorm, p.printLevel, p.outputFile = n, nev, 1e-9, 4.0, 0, None
    p.matrixMatvec = C.cast(cmv, C.c_void_p); p.applyPreconditioner = C.cast(cpc, C.c_void_p)
    p.correctionParams.precondition = 1
    p.initBasisMode = F.primme_init_user; p.initSize = 1
    lib.primme_set_method(F.PRIMME_GD_plusK, C.byref(p))
    evecs = np.zeros((nev, n)); evecs[0] = problems.start_vector(n)
    evals, rn = np.zeros(nev), np.zeros(nev)
    lib.dprimme.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(F.PrimmeParams)]
    ret = lib.dprimme(evals.ctypes.data, evecs.ctypes.data, rn.ctypes.data, C.byref(p))
    assert ret == 0 and p.initSize == nev and seen["user_struct"]
    assert np.max(np.abs(evals - problems.laplacian_eigenvalues((n,), nev))) <= 1e-10 * 4.0
    assert seen["mv"] == p.stats.numMatvecs and seen["pc"] == p.stats.numPreconds > 0
    A = np.zeros((n, n)); A[np.repeat(np.arange(n), np.diff(rp)), ci] = va
    X = evecs.T
    assert np.all(np.linalg.norm(A @ X - X * evals, axis=0) <= 1e-9 * 4.0 * 1.05)
    assert C.cast(p.matrixMatvec, C.c_void_p).value == C.cast(cmv, C.c_void_p).value and not p.queue   # caller's struct restored
    # the device-pointer entry on the same problem, ready-made operator: same history
    g = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", numEvals=nev, eps=1e-9, aNorm=4.0, precond="jacobi",
              method="GD_plusK", v0=problems.start_vector(n))
    assert g.ret == 0 and g.stats["numOuterIterations"] == p.stats.numOuterIterations and g.stats["numMatvecs"] == p.stats.numMatvecs
    assert np.max(np.abs(g.evals - evals)) <= 1e-13


@pytest.mark.parametrize("kw", [dict(numEvals=10, maxBlockSize=8), dict(numEvals=6, target="largest", maxBlockSize=8), dict(numEvals=9, maxBlockSize=3),
                                dict(numEvals=8, maxBlockSize=8, locking=1)])
def test_block_qmr_step_with_one_synchronisation_keeps_the_history(built, kw, monkeypatch):
    """Round 5: block JDQMR with the library's Jacobi preconditioner runs every inner step as three launches and ONE host wait — the
    step's alpha, gamma, eta, beta are evaluated on the device, in the prologue of the launch that applies them
    (hipk_axpy_proj_dot_jacobi_dev, hipk_qmr_update_dir_dev; oracle/hipk_cpu.c restates them), the host evaluates the same
    expressions afterwards for its stopping tests.  Same roundings on both sides: the history is the three-wait sequence's
    (PRIMME_AMD_QMR_THREE_WAITS=1) bit for bit — iterations, operator and preconditioner applications, eigenvalues, residual norms."""
    import ctypes as C
    rp, ci, va, n = problems.laplacian_csr((22, 23, 13))
    v0 = np.random.default_rng(7).standard_normal((n, kw["maxBlockSize"]))
    lib = checkers.load_hostcheck()
    lib.primme_amd_qmr_step_stats(None, None)
    runs = []
    for three in (False, True):
        if three: monkeypatch.setenv("PRIMME_AMD_QMR_THREE_WAITS", "1")
        else: monkeypatch.delenv("PRIMME_AMD_QMR_THREE_WAITS", raising=False)
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", method="JDQMR", eps=1e-9, aNorm=12.0, v0=v0, precond=("jacobi", 0.0), **kw)
        assert r.ret == 0
        qs = (C.c_long * 2)()
        lib.primme_amd_qmr_step_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.primme_amd_qmr_step_stats(C.cast(qs, C.POINTER(C.c_long)), C.cast(C.byref(qs, C.sizeof(C.c_long)), C.POINTER(C.c_long)))
        assert qs[0] > 20 and (qs[1] == 0 if three else qs[1] >= 0.8 * qs[0]), (three, qs[0], qs[1])     # (the last step of an inner solve keeps the old form)
        runs.append((r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numPreconds"], r.evals.tobytes(), r.resNorms.tobytes()))
    assert runs[0] == runs[1]
    ex = problems.laplacian_eigenvalues((22, 23, 13), kw["numEvals"])
    want = ex if kw.get("target") != "largest" else 12.0 - ex        # the spectrum of the 3-D stencil is symmetric about 6
    assert np.max(np.abs(np.sort(np.frombuffer(runs[0][3])) - np.sort(want))) <= 1e-8 * 12.0
