"""N > 1 path on CPU: world_size 2, gloo, row-partitioned solve through the reference's
globalSumReal contract (host buffers), compared with the single-rank solve of the same matrix."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from primme_amd import problems
from checkers import eigsh, Operator

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(case, tmp_path, world=2):
    port = _free_port()
    out = str(tmp_path / f"res_{case}")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker.py"), str(r), str(world), str(port), case, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def test_block_diagonal_two_ranks(built, tmp_path):
    res = _launch("blockdiag", tmp_path)
    dims = (15, 16)
    rp, ci, va, n0 = problems.laplacian_csr(dims)
    rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 2, scale_fn=lambda t: 1.0 + 0.37 * t)
    n = 2 * n0
    single = eigsh(Operator(n, csr=(rpt, cit, vat)), backend="hostcheck", numEvals=6, eps=1e-10, aNorm=8.0 * 1.37,
                   v0=problems.start_vector(n))
    ex = np.sort(np.concatenate([problems.laplacian_eigenvalues(dims, 6), 1.37 * problems.laplacian_eigenvalues(dims, 6)]))[:6]
    for r in res:
        assert r["ret"] == 0
        assert np.max(np.abs(np.array(r["evals"]) - ex)) <= 1e-10 * 8 * 1.37
        assert np.max(np.abs(np.array(r["evals"]) - single.evals)) <= 1e-10 * 8 * 1.37
        assert r["numGlobalSum"] > 0
    # every rank computed the identical small problem: bitwise equal eigenvalues, same iteration count
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"]
    # the local slabs together are unit vectors
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - 6.0) < 1e-8
    assert abs(res[0]["its"] - single.stats["numOuterIterations"]) <= 0.05 * single.stats["numOuterIterations"]


@pytest.mark.parametrize("case", ["blockdiag_mass", "blockdiag_mass_jdqmr"])
def test_generalized_problem_two_ranks(built, tmp_path, case):
    """A x = lambda B x with the rows over two ranks (round 6): GD+k through the tracked-Gram path and JDQMR with its projectors on
    B Q / B x, every inner product through globalSumReal — scipy's dense truth, the one-rank solve of the same problem, identical
    bits on both ranks, B-orthonormal slabs."""
    import scipy.linalg as sl
    import scipy.sparse as sp
    res = _launch(case, tmp_path)
    dims = (15, 16)
    rp, ci, va, n0 = problems.laplacian_csr(dims)
    rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 2, scale_fn=lambda t: 1.0 + 0.37 * t)
    brp, bci, bva = problems.mass_matrix_csr(n0)
    brpt, bcit, bvat = problems.tile_block_diagonal(brp, bci, bva, 2, scale_fn=lambda t: 1.0 + 0.11 * t)
    n = 2 * n0
    kw = dict(method="JDQMR", precond="jacobi", locking=1) if case.endswith("_jdqmr") else dict(method="GD_plusK")
    single = eigsh(Operator(n, csr=(rpt, cit, vat)), backend="hostcheck", mass=Operator(n, csr=(brpt, bcit, bvat)), numEvals=5, eps=1e-9,
                   aNorm=8.0 * 1.37, v0=problems.start_vector(n), **kw)
    w = sl.eigh(sp.csr_matrix((vat, cit, rpt), shape=(n, n)).toarray(), sp.csr_matrix((bvat, bcit, brpt), shape=(n, n)).toarray(), eigvals_only=True)[:5]
    assert single.ret == 0
    for r in res:
        assert r["ret"] == 0 and r["numGlobalSum"] > 0
        assert np.max(np.abs(np.sort(r["evals"]) - w)) <= 1e-9 * 8 * 1.37
        assert np.max(np.abs(np.sort(r["evals"]) - np.sort(single.evals))) <= 1e-9 * 8 * 1.37
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"]
    assert abs(res[0]["its"] - single.stats["numOuterIterations"]) <= max(2, 0.1 * single.stats["numOuterIterations"])


def test_halo_exchange_two_ranks(built, tmp_path):
    res = _launch("halo", tmp_path)
    dims = (20, 22)
    ex = problems.laplacian_eigenvalues(dims, 5)
    for r in res:
        assert r["ret"] == 0
        assert np.max(np.abs(np.array(r["evals"]) - ex)) <= 1e-10 * 8
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - 5.0) < 1e-8


def test_svds_two_ranks(built, tmp_path):
    """Singular values with A split by rows over two ranks (all-gather / reduce-scatter matvec through
    user callbacks): same triplets as the dense truth, identical on both ranks."""
    res = _launch("svds", tmp_path)
    m, n, k = 600, 200, 4
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    A = np.zeros((m, n))
    A[np.repeat(np.arange(m), np.diff(rp)), ci] = va
    s = np.linalg.svd(A, compute_uv=False)
    for r in res:
        assert r["ret"] == 0
        assert np.max(np.abs(np.array(r["evals"]) - s[:k])) <= 1e-10 * s[0]
        assert np.all(np.array(r["resNorms"]) <= 1e-10 * r["aNorm"] * (1 + 1e-6))
        assert r["numGlobalSum"] > 0
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - k) < 1e-8
    assert abs(res[0]["u_norm2"] + res[1]["u_norm2"] - k) < 1e-8


@pytest.mark.parametrize("case", ["hermitian", "hermitian_blk4", "hermitian_jdqmr", "hermitian_harmonic", "hermitian_refined"])
def test_hermitian_two_ranks(built, tmp_path, case):
    """hip_zprimme with the rows of a complex Hermitian matrix on two ranks: the native complex solve (GD+k, the block
    iteration of configs[3], JDQMR, harmonic and refined extraction) reduces its (re, im) inner products through
    globalSumReal; both ranks return the same eigenvalues bit for bit."""
    res = _launch(case, tmp_path)
    nloc = 150
    rp, ci, va = problems.hermitian_banded_csr(nloc)
    A1 = np.zeros((nloc, nloc), dtype=np.complex128)
    A1[np.repeat(np.arange(nloc), np.diff(rp)), ci] = va
    assert np.allclose(A1, A1.conj().T)
    wall = np.concatenate([np.linalg.eigvalsh(A1), 1.21 * np.linalg.eigvalsh(A1)])
    interior = case in ("hermitian_harmonic", "hermitian_refined")
    w = wall[np.argsort(np.abs(wall - 2.5))][:4] if interior else np.sort(wall)[::-1][:4]
    for r in res:
        assert r["ret"] == 0
        got = np.array(r["evals"])
        if interior: got, w = np.sort(got), np.sort(w)
        assert np.max(np.abs(got - w)) <= (1e-7 if interior else 1e-10) * 1.21 * 4
        assert r["numGlobalSum"] > 0
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - 4.0) < 1e-8


def test_hermitian_device_communicator_two_ranks(built, tmp_path):
    """BASELINE configs[3] in small the way the GPUs run it: native complex panels, the library's complex CSR operator,
    block size 4, and the library's communicator reducing the (re, im) partial sums in the stream (stand-in: gloo)."""
    res = _launch("zdevcomm", tmp_path)
    nloc = 150
    rp, ci, va = problems.hermitian_banded_csr(nloc)
    A1 = np.zeros((nloc, nloc), dtype=np.complex128)
    A1[np.repeat(np.arange(nloc), np.diff(rp)), ci] = va
    w = np.sort(np.concatenate([np.linalg.eigvalsh(A1), 1.21 * np.linalg.eigvalsh(A1)]))[::-1][:6]
    for r in res:
        assert r["ret"] == 0 and r["allreduces"] > 0
        assert np.max(np.abs(np.array(r["evals"]) - w)) <= 1e-9 * 1.21 * 4
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["allreduces"] == res[1]["allreduces"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - 6.0) < 1e-7


@pytest.mark.parametrize("case,nev", [("devcomm_lock", 10), ("devcomm_soft", 4)])
def test_device_communicator_path_two_ranks(built, tmp_path, case, nev):
    """The code path of a multi-GPU run -- the library's own operator AND communicator: all-reduces inside the stream of
    launches, |t|^2 and t'At in one all-reduce, fused and speculative restart working on reduced overlaps -- with
    world_size 2 on CPU: the communicator is the stand-in of oracle/hostcheck_glue.c whose all-reduce is gloo."""
    res = _launch(case, tmp_path)
    dims = (15, 16)
    rp, ci, va, n0 = problems.laplacian_csr(dims)
    rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 2, scale_fn=lambda t: 1.0 + 0.37 * t)
    n = 2 * n0
    single = eigsh(Operator(n, csr=(rpt, cit, vat)), backend="hostcheck", numEvals=nev, eps=1e-10, aNorm=8.0 * 1.37,
                   v0=problems.start_vector(n))
    ex = np.sort(np.concatenate([problems.laplacian_eigenvalues(dims, nev), 1.37 * problems.laplacian_eigenvalues(dims, nev)]))[:nev]
    for r in res:
        assert r["ret"] == 0 and r["locking"] == (1 if case == "devcomm_lock" else 0)
        assert np.max(np.abs(np.array(r["evals"]) - ex)) <= 1e-10 * 8 * 1.37
        assert np.max(np.abs(np.array(r["evals"]) - single.evals)) <= 1e-10 * 8 * 1.37
        assert np.all(np.array(r["resNorms"]) <= 1e-10 * 8 * 1.37 * 1.001)
        # the single-rank launch structure survives: one-launch tail in (almost) every iteration, the check at the full
        # basis is the restart pass, panel products only around locked pairs; two all-reduces per iteration + change
        assert r["fused_tail"] >= r["its"] - 20 and r["ritz_ov"] >= r["restarts"] - 12
        assert r["dots"] <= r["restarts"] + 60
        assert 2 * r["its"] <= r["allreduces"] <= 2 * r["its"] + 3 * r["restarts"] + 80
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["matvecs"] == res[1]["matvecs"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - nev) < 1e-8
    assert abs(res[0]["its"] - single.stats["numOuterIterations"]) <= max(3, 0.05 * single.stats["numOuterIterations"])


def test_device_communicator_path_with_halo_two_ranks(built, tmp_path):
    """The same path on ONE Laplacian split by rows (the bench's partition): the operator exchanges halo rows and the
    fused tail (hipk_csr_matvec_scaled: scale + A t + t'At) runs with halo buffers — the combination whose one-GPU
    kernel test failed once in the round-2 driver run.  Eigenvalues against the analytic spectrum and the one-rank
    solve, identical bits on both ranks, the one-launch tail in (almost) every iteration."""
    res = _launch("devcomm_halo", tmp_path)
    dims = (24, 22)
    rp, ci, va, n = problems.laplacian_csr(dims)
    single = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", numEvals=6, eps=1e-10, aNorm=8.0, v0=problems.start_vector(n))
    ex = problems.laplacian_eigenvalues(dims, 6)
    for r in res:
        assert r["ret"] == 0
        assert np.max(np.abs(np.sort(r["evals"]) - ex)) <= 1e-10 * 8
        assert np.max(np.abs(np.array(r["evals"]) - single.evals)) <= 1e-10 * 8
        assert np.all(np.array(r["resNorms"]) <= 1e-10 * 8 * 1.001)
        assert r["fused_tail"] >= r["its"] - 20
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["matvecs"] == res[1]["matvecs"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - 6) < 1e-8
    assert abs(res[0]["its"] - single.stats["numOuterIterations"]) <= max(3, 0.05 * single.stats["numOuterIterations"])


@pytest.mark.parametrize("case,nev", [("devcomm_lock_xr", 10), ("devcomm_soft_xr", 4), ("devcomm_halo_xr", 6)])
def test_peer_to_peer_second_stage_and_iterations_enqueued_ahead_two_ranks(built, tmp_path, case, nev):
    """Round 5.  The host logic of ranks on the MAILBOX transport (comm_ipc.hip), world_size 2 on CPU: every reduction of the
    block-size-1 iteration is exchanged inside the second stage of the launch that forms it (hipk_xreduce_arm; stand-in:
    oracle/hostcheck_glue.c), so the global sums are in "HBM" without the host and the NEXT iteration is enqueued before the host
    has seen the current one (eigs_conv.c: pa_prelaunch_next), on both ranks alike.  Same iteration / matvec / restart counts as
    the same solve with separate all-reduces (the RCCL form), identical bits on both ranks, most iterations adopted."""
    res = _launch(case, tmp_path)
    plain = _launch(case[:-3], tmp_path)
    for r, q in zip(res, plain):
        assert r["ret"] == 0 and q["ret"] == 0
        assert (r["its"], r["matvecs"], r["restarts"]) == (q["its"], q["matvecs"], q["restarts"])
        assert np.max(np.abs(np.array(r["evals"]) - np.array(q["evals"]))) <= 1e-12 * 8 * 1.37
        assert np.all(np.array(r["resNorms"]) <= 1e-10 * 8 * 1.37 * 1.001)
        # three exchanges per iteration, each inside the launch that forms the sums; iterations enqueued ahead: all but the ones
        # around restarts and converged pairs, and nearly all of them adopted
        assert r["fused_exchanges"] >= 3 * (r["its"] - r["restarts"]) - 60 and q["fused_exchanges"] == 0
        assert r["ahead"] >= r["its"] - r["restarts"] - 40 and r["adopted"] >= r["ahead"] - 12 and q["ahead"] == 0
        assert r["fused_tail"] >= r["its"] - 20
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["matvecs"] == res[1]["matvecs"]
    assert res[0]["adopted"] == res[1]["adopted"] and res[0]["ahead"] == res[1]["ahead"] and res[0]["allreduces"] == res[1]["allreduces"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - nev) < 1e-8


def test_block_qmr_one_synchronisation_two_ranks(built, tmp_path):
    """Round 5: the block QMR step with one host synchronisation (eigs_jd.c) on the device-communicator path, world_size 2 on CPU:
    the three inner products / the two reductions of the step are all-reduced inside the stream of launches and the step's scalar
    recurrences are evaluated from the GLOBAL sums, next to the launches that apply them and — afterwards — on every host.  Same
    history as the three-wait sequence bit for bit, identical bits on both ranks, the one-rank solve's eigenvalues."""
    res = _launch("devcomm_jdqmr", tmp_path)
    old = _launch("devcomm_jdqmr3", tmp_path)
    dims = (15, 16)
    ex = np.sort(np.concatenate([problems.laplacian_eigenvalues(dims, 6), 1.37 * problems.laplacian_eigenvalues(dims, 6)]))[:6]
    for r, q in zip(res, old):
        assert r["ret"] == 0 and q["ret"] == 0
        assert (r["its"], r["matvecs"], r["evals"], r["resNorms"]) == (q["its"], q["matvecs"], q["evals"], q["resNorms"])
        assert np.max(np.abs(np.sort(r["evals"]) - ex)) <= 1e-9 * 8 * 1.37
        assert r["allreduces"] > 0
        assert r["qmr_steps"] == q["qmr_steps"] > 50 and r["qmr_steps_one_wait"] >= 0.8 * r["qmr_steps"] and q["qmr_steps_one_wait"] == 0
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["matvecs"] == res[1]["matvecs"]


def test_complex_svds_native_two_ranks(built, tmp_path):
    """Round 5: hip_zprimme_svds on two ranks — the native complex singular value front end (csrc/svds_main.c on complex panels over
    the native complex eigensolver) with A split by rows and all-gather / reduce-scatter callbacks on complex blocks: the dense
    truth's triplets, identical on both ranks, unit-norm complex vectors."""
    res = _launch("svds_z", tmp_path)
    m, n, k = 600, 200, 4
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    vz = va.astype(np.complex128) * np.exp(1j * np.random.default_rng(5).uniform(0, 2 * np.pi, size=len(va)))
    A = np.zeros((m, n), dtype=np.complex128)
    A[np.repeat(np.arange(m), np.diff(rp)), ci] = vz
    s = np.linalg.svd(A, compute_uv=False)
    for r in res:
        assert r["ret"] == 0
        assert np.max(np.abs(np.array(r["evals"]) - s[:k])) <= 1e-10 * s[0]
        assert np.all(np.array(r["resNorms"]) <= 1e-10 * r["aNorm"] * (1 + 1e-6))
        assert r["numGlobalSum"] > 0
    assert res[0]["evals"] == res[1]["evals"] and res[0]["its"] == res[1]["its"] and res[0]["matvecs"] == res[1]["matvecs"]
    assert abs(res[0]["evecs_norm2"] + res[1]["evecs_norm2"] - k) < 1e-8
    assert abs(res[0]["u_norm2"] + res[1]["u_norm2"] - k) < 1e-8
