"""The row-partitioned device path with MORE THAN ONE RANK on real hardware: 2, 4 and 8 processes share the one
GPU of the test box, the communicator runs on the peer-to-peer transport (PRIMME_AMD_COMM=ipc, csrc/comm_ipc.hip:
mailboxes in device memory exported with hipIpcGetMemHandle and written by the peers directly), rows are split as
in the reference's MPI example (examples/ex_eigs_mpi.c:100-123), globalSumReal is the library's callback
(examples/ex_eigs_mpi.c:209-218 wraps MPI_Allreduce there).  RCCL refuses two ranks on one device; the mailboxes
do not care whether a peer is another GPU over xGMI or another process on the same GPU, so every kernel, flag
and generation hand-off of the multi-GPU path executes here.

Checked: every rank returns identical bits (they solve the same projected problem from the same rank-ordered
sums); the answers equal the one-rank solve of the same problem; the collectives themselves against numpy."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from primme_amd import problems
from checkers import eigsh, Operator, svds

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORLDS = [2, 4, 8]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(case, world, tmp_path, extra_env=None):
    port = _free_port()
    out = str(tmp_path / f"res_{case}_{world}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PRIMME_AMD_COMM="ipc", PRIMME_AMD_IPC_DEVICE_TIMEOUT_S="120")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker_gpu.py"), str(r), str(world), str(port), case, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    try:
        outs = [p.communicate(timeout=900)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    for r in res:
        assert r["transport"] == "ipc"
        assert r["ret"] == 0, r
        assert r["numGlobalSum"] > 0
        assert r["evals"] == res[0]["evals"] and r["its"] == res[0]["its"] and r["resNorms"] == res[0]["resNorms"]   # identical bits
    return res


@pytest.mark.parametrize("world", WORLDS)
def test_ipc_collectives(built, tmp_path, world):
    """all-reduce (sizes across the slot and block boundaries), neighbour halo, all-gather, reduce-scatter, the
    integer exchange: against numpy; the reduction results are bitwise the same on every rank."""
    res = _launch("comm_ops", world, tmp_path)
    for r in res:
        assert r["digest"] == res[0]["digest"] and r["selftest"] == 0
    lat = {"world": world, "selftest_allreduce_us": [round(r["selftest_allreduce_us"], 1) for r in res], "allreduce8_sync_us": [round(r["allreduce8_sync_us"], 1) for r in res],
           "allreduce8_chain_us": [round(r["allreduce8_chain_us"], 1) for r in res]}
    print("IPC_LATENCY", json.dumps(lat))
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "ipc_allreduce_latency.jsonl"), "a") as f:
        f.write(json.dumps(lat) + "\n")


def test_ipc_tag_sequence_wraps_around(built, tmp_path):
    """The 32-bit tag sequence of the mailbox reductions started 60 reductions below 2^32 (PRIMME_AMD_IPC_SEQ0): the
    collectives cross the wrap-around — every rank drains, meets the others and clears its granule area there
    (csrc/comm_ipc.hip: ipc_seq_wrap), the sequence continues at 2 with the generations alternating — and every sum
    is still the rank-ordered one on every rank, the library's self-test included."""
    res = _launch("comm_ops", 2, tmp_path, extra_env={"PRIMME_AMD_IPC_SEQ0": str(2 ** 32 - 60)})
    for r in res:
        assert r["digest"] == res[0]["digest"] and r["selftest"] == 0


@pytest.mark.parametrize("world", WORLDS)
def test_ipc_configs1_small(built, tmp_path, world):
    """BASELINE configs[1] in small (3-D 7-point Laplacian, 10 smallest, GD+k, block size 1): the fused
    one-synchronisation iteration with its halo and its merged reductions, rows over `world` processes."""
    res = _launch("lap3d_small", world, tmp_path)
    dims = (40, 41, 42)
    rp, ci, va, n = problems.laplacian_csr(dims)
    one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", numEvals=10, eps=1e-8, aNorm=12.0, method="GD_plusK", v0=problems.start_vector(n))
    assert one.ret == 0
    ex = problems.laplacian_eigenvalues(dims, 10)
    ev = np.array(res[0]["evals"])
    assert np.max(np.abs(np.sort(ev) - ex)) <= 1e-8 * 12.0
    assert np.max(np.abs(ev - one.evals)) <= 1e-10 * 12.0                       # north_star: 1e-10 relative to |A|
    assert np.all(np.array(res[0]["resNorms"]) <= 1e-8 * 12.0 * (1 + 1e-6))
    assert abs(sum(r["evecs_norm2"] for r in res) - 10.0) < 1e-8
    assert abs(res[0]["its"] - one.stats["numOuterIterations"]) <= max(3, 0.05 * one.stats["numOuterIterations"])
    assert res[0]["numGlobalSum"] >= res[0]["its"]
    # round 5: on the mailboxes the global sums are in HBM without the host, so the next iteration is enqueued before the host has
    # seen the current one on every rank alike (eigs_conv.c: pa_prelaunch_next) — most iterations, same decisions on all ranks
    assert res[0]["adopted"] >= 0.6 * res[0]["its"] and res[0]["ahead"] - res[0]["adopted"] <= 40
    assert all(r["adopted"] == res[0]["adopted"] and r["ahead"] == res[0]["ahead"] for r in res)


@pytest.mark.parametrize("world,case", [(2, "halo"), (4, "halo"), (8, "halo"), (4, "halo_block")])
def test_ipc_halo_laplacian(built, tmp_path, world, case):
    res = _launch(case, world, tmp_path)
    dims = (24, 25, 26)
    ex = problems.laplacian_eigenvalues(dims, 6)
    assert np.max(np.abs(np.sort(res[0]["evals"]) - ex)) <= 1e-10 * 12.0
    assert np.all(np.array(res[0]["resNorms"]) <= 1e-10 * 12.0 * (1 + 1e-6))
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8
    if case == "halo":
        rp, ci, va, n = problems.laplacian_csr(dims)
        one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", numEvals=6, eps=1e-10, aNorm=12.0, v0=problems.start_vector(n))
        assert np.max(np.abs(np.array(res[0]["evals"]) - one.evals)) <= 1e-10 * 12.0
        assert abs(res[0]["its"] - one.stats["numOuterIterations"]) <= max(2, 0.03 * one.stats["numOuterIterations"])
        assert res[0]["adopted"] >= 0.6 * res[0]["its"] and all(r["adopted"] == res[0]["adopted"] for r in res)


@pytest.mark.parametrize("world,case", [(2, "halo_mass"), (4, "halo_mass"), (2, "halo_mass_jdqmr"), (4, "halo_mass_jdqmr")])
def test_ipc_generalized_problem(built, tmp_path, world, case):
    """A x = lambda B x with the rows over `world` processes on the mailboxes (round 6): A and B are two ready-made operators with
    their own halo exchanges; GD+k through the tracked-Gram path and JDQMR with its projectors on B Q / B x.  Against the one-rank
    solve on the device: the same eigenvalues to 1e-10 |A|, the same iteration count to 5 % (15 % for the inner-outer method),
    identical bits on all ranks."""
    res = _launch(case, world, tmp_path)
    dims = (24, 25, 26)
    rp, ci, va, n = problems.laplacian_csr(dims)
    brp, bci, bva = problems.mass_matrix_csr(n)
    kw = dict(method="JDQMR", precond="jacobi", locking=1) if case.endswith("_jdqmr") else dict(method="GD_plusK")
    one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", mass=Operator(n, csr=(brp, bci, bva)), numEvals=5, eps=1e-9, aNorm=12.0,
                v0=problems.start_vector(n), **kw)
    assert one.ret == 0
    for r in res:
        assert r["ret"] == 0 and r["evals"] == res[0]["evals"] and r["its"] == res[0]["its"]
    assert np.max(np.abs(np.sort(res[0]["evals"]) - np.sort(one.evals))) <= 1e-10 * 12.0
    tol = 0.15 if case.endswith("_jdqmr") else 0.05
    assert abs(res[0]["its"] - one.stats["numOuterIterations"]) <= max(2, tol * one.stats["numOuterIterations"])
    # B-orthonormal Ritz vectors: the Euclidean norms of the slabs add up to x'x = 1 / (x'Bx / x'x) per vector, between 1/|B| and |B^-1|
    tot = sum(r["evecs_norm2"] for r in res)
    assert 5.0 / 1.8 < tot < 5.0 / 0.7


@pytest.mark.parametrize("world", WORLDS)
def test_ipc_configs3_small(built, tmp_path, world):
    """BASELINE configs[3] in small: complex Hermitian banded, block size 4, 6 largest, rows split."""
    res = _launch("hermitian_b4", world, tmp_path)
    n = 4000 * world
    rp, ci, va = problems.hermitian_banded_csr(n)
    one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", dtype=np.complex128, numEvals=6, target="largest", eps=1e-10,
                iseed=(5, 1, 2, 3), maxBlockSize=4, method="GD_plusK")
    assert one.ret == 0
    assert np.max(np.abs(np.array(res[0]["evals"]) - one.evals)) <= 1e-10 * one.params["aNorm"]
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8


@pytest.mark.parametrize("world", [2])
def test_ipc_allgather_and_svds(built, tmp_path, world):
    """the bulk window: unstructured columns (whole-vector gather per block) and the singular value operator's
    all-gather / reduce-scatter pair"""
    res = _launch("allgather", world, tmp_path)
    import scipy.sparse as sp
    dims = (64, 8 * world)
    rp, ci, va, n = problems.laplacian_csr(dims)
    A = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
    h = n // 2
    for i in range(0, h, 5):
        A[i, i + h] = 0.25; A[i + h, i] = 0.25
    w = np.linalg.eigvalsh(A.toarray())[:4]
    assert np.max(np.abs(np.array(res[0]["evals"]) - w)) <= 1e-10 * 8.5
    res = _launch("svds", world, tmp_path)
    m, n, k = 4000 * world, 500 * world, 5
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    one = svds(m, n, (rp, ci, va), numSvals=k, eps=1e-10, methodStage1="GD_plusK", backend="hip", maxBlockSize=2)
    assert one.ret == 0
    assert np.max(np.abs(np.array(res[0]["evals"]) - one.svals)) <= 1e-10 * one.svals[0]
    assert abs(sum(r["evecs_norm2"] for r in res) - k) < 1e-8 and abs(sum(r["u_norm2"] for r in res) - k) < 1e-8


def test_ipc_transport_agrees_with_separate_launches(built, tmp_path):
    """the fused second stage (finalize + exchange + publication in one launch) against the same transport with the
    reduction as a launch of its own: same sums, same history"""
    a = _launch("lap3d_small", 2, tmp_path)
    b = _launch("lap3d_small", 2, tmp_path, extra_env={"PRIMME_AMD_NO_XREDUCE": "1"})
    assert a[0]["its"] == b[0]["its"] and a[0]["matvecs"] == b[0]["matvecs"]
    assert np.max(np.abs(np.array(a[0]["evals"]) - np.array(b[0]["evals"]))) <= 1e-12 * 12.0


def test_ipc_iterations_enqueued_ahead_agree_with_the_plain_sequence(built, tmp_path):
    """Round 5: the row-partitioned run with the next iteration enqueued before the host has seen the current one (default on the
    mailboxes) against the same run with every iteration launched after the host's own Rayleigh-Ritz solve
    (PRIMME_AMD_NO_PRELAUNCH=1): same history, same pairs."""
    for case in ("lap3d_small", "halo"):
        a = _launch(case, 2, tmp_path)
        b = _launch(case, 2, tmp_path, extra_env={"PRIMME_AMD_NO_PRELAUNCH": "1"})
        assert a[0]["adopted"] > 0 and b[0]["ahead"] == 0
        assert a[0]["its"] == b[0]["its"] and a[0]["matvecs"] == b[0]["matvecs"]
        assert np.max(np.abs(np.array(a[0]["evals"]) - np.array(b[0]["evals"]))) <= 1e-12 * 12.0
        assert np.max(np.abs(np.array(a[0]["resNorms"]) - np.array(b[0]["resNorms"]))) <= 1e-10 * 12.0
