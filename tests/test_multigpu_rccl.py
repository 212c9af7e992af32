"""The row-partitioned solve over RCCL on however many GPUs the box has (one process per GPU): the
library's own communicator — in-stream ncclAllReduce of the <= 4 KB inner-product panels, neighbour
halo exchange (grouped ncclSend / ncclRecv), grouped all-gather and reduce-scatter — against the
single-GPU solve of the same problem.  Mirrors tests/test_multirank_gloo.py (host callbacks, CPU).

With ONE visible GPU the same worker runs with world size 1 (every collective still goes through
RCCL, as tests/test_comm_gpu.py does); with >= 2 it is the real thing: results must be bitwise
identical on every rank (each rank solves the same small projected problem from the same reduced
panels) and equal to the one-rank answer."""
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


def _world():
    import torch
    return max(1, min(torch.cuda.device_count(), 8))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(case, tmp_path, transport="rccl", strict=False):
    """transport: PRIMME_AMD_COMM of the ranks.  This module is about the RCCL transport (the default argument); one case
    also runs under "auto" (peer-to-peer mailboxes + RCCL when every rank has its own GPU, mailboxes alone with one rank)."""
    world = _world()
    port = _free_port()
    out = str(tmp_path / f"res_{case}_{transport}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PRIMME_AMD_COMM=transport)
    if strict:
        env["PRIMME_AMD_COMM_STRICT"] = "1"      # mailboxes that do not map across the devices FAIL the create instead of falling back to RCCL
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker_gpu.py"), str(r), str(world), str(port), case, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [json.load(open(f"{out}.{r}")) for r in range(world)]
    for r in res:
        assert r["ret"] == 0, r
        assert r["numGlobalSum"] > 0
        assert r["transport"] == ("rccl" if transport == "rccl" else ("hybrid" if world > 1 else "ipc")), r["transport"]
        # every rank reduced the same panels and solved the same projected problem: identical bits
        assert r["evals"] == res[0]["evals"] and r["its"] == res[0]["its"] and r["resNorms"] == res[0]["resNorms"]
    return world, res


@pytest.mark.parametrize("case", ["halo", "halo_block"])
def test_rccl_halo_stencil(built, tmp_path, case):
    world, res = _launch(case, tmp_path)
    dims = (24, 25, 26)
    ex = problems.laplacian_eigenvalues(dims, 6)
    assert np.max(np.abs(np.sort(res[0]["evals"]) - ex)) <= 1e-10 * 12.0
    assert np.all(np.array(res[0]["resNorms"]) <= 1e-10 * 12.0 * (1 + 1e-6))
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8           # the slabs together are unit vectors
    if case == "halo":
        rp, ci, va, n = problems.laplacian_csr(dims)
        one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", numEvals=6, eps=1e-10, aNorm=12.0, v0=problems.start_vector(n))
        assert np.max(np.abs(np.array(res[0]["evals"]) - one.evals)) <= 1e-10 * 12.0
        assert abs(res[0]["its"] - one.stats["numOuterIterations"]) <= max(2, 0.03 * one.stats["numOuterIterations"])


@pytest.mark.parametrize("case,dims,nev,aN", [("config2_full", (125, 126, 127), 10, 12.0), ("lap2d_10m", (3162, 3163), 2, 8.0)])
def test_rccl_bench_workloads_under_the_bench_partition(built, tmp_path, case, dims, nev, aN):
    """What `bench.py --gpus N` runs, checked instead of timed: BASELINE configs[1] at full size and the north-star
    10 M-row Laplacian, rows split over every visible GPU, against the analytic spectrum; identical bits on all ranks
    (asserted in _launch); the slabs of the returned vectors are together unit vectors."""
    world, res = _launch(case, tmp_path)
    ex = problems.laplacian_eigenvalues(dims, nev)
    assert np.max(np.abs(np.sort(res[0]["evals"]) - ex)) <= 1e-8 * aN
    assert np.all(np.array(res[0]["resNorms"]) <= 1e-8 * aN * (1 + 1e-6))
    assert abs(sum(r["evecs_norm2"] for r in res) - nev) < 1e-6
    assert res[0]["numGlobalSum"] >= res[0]["its"]          # every outer iteration reduced something across ranks


def test_auto_transport_halo_stencil(built, tmp_path):
    """the same halo case under PRIMME_AMD_COMM=auto: with one GPU per rank this is the mailboxes for reductions and halos
    next to RCCL for the bulk collectives — the form `bench.py --gpus N` runs by default"""
    world, res = _launch("halo", tmp_path, transport="auto")
    ex = problems.laplacian_eigenvalues((24, 25, 26), 6)
    assert np.max(np.abs(np.sort(res[0]["evals"]) - ex)) <= 1e-10 * 12.0
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8


@pytest.mark.parametrize("transport", ["auto", "rccl"])
def test_both_transports_strict_at_the_full_device_count(built, tmp_path, transport):
    """First contact with distinct GPUs made boring (VERDICT r05 Next #7): at world = device_count() the halo case must pass on
    BOTH transports, and under PRIMME_AMD_COMM_STRICT=1 the auto transport must really be the peer-to-peer mailboxes — if they
    do not map across the devices the communicator fails to come up (the ranks exit non-zero and this test fails) instead of
    falling back to RCCL silently; _launch asserts the transport every rank reports ("hybrid" with more than one device, "ipc"
    with one, "rccl" when asked for)."""
    world, res = _launch("halo", tmp_path, transport=transport, strict=True)
    ex = problems.laplacian_eigenvalues((24, 25, 26), 6)
    assert np.max(np.abs(np.sort(res[0]["evals"]) - ex)) <= 1e-10 * 12.0
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8


def test_rccl_block_diagonal(built, tmp_path):
    world, res = _launch("blockdiag", tmp_path)
    dims = (40, 41)
    scales = [1.0 + 0.37 * t / max(world - 1, 1) for t in range(world)]
    ex = np.sort(np.concatenate([s * problems.laplacian_eigenvalues(dims, 6) for s in scales]))[:6]
    assert np.max(np.abs(np.array(res[0]["evals"]) - ex)) <= 1e-10 * 8 * 1.37
    assert abs(sum(r["evecs_norm2"] for r in res) - 6.0) < 1e-8


def test_rccl_allgather_unstructured_columns(built, tmp_path):
    world, res = _launch("allgather", tmp_path)
    import scipy.sparse as sp
    dims = (64, 8 * world)
    rp, ci, va, n = problems.laplacian_csr(dims)
    A = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
    h = n // 2
    for i in range(0, h, 5):
        A[i, i + h] = 0.25; A[i + h, i] = 0.25
    w = np.linalg.eigvalsh(A.toarray())[:4]
    assert np.max(np.abs(np.array(res[0]["evals"]) - w)) <= 1e-10 * 8.5
    assert abs(sum(r["evecs_norm2"] for r in res) - 4.0) < 1e-8


def test_rccl_hermitian_rows_split(built, tmp_path):
    world, res = _launch("hermitian", tmp_path)
    n = 3000 * world
    rp, ci, va = problems.hermitian_banded_csr(n)
    one = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", dtype=np.complex128, numEvals=4, target="largest", eps=1e-10,
                iseed=(5, 1, 2, 3), maxBlockSize=2)
    assert one.ret == 0
    assert np.max(np.abs(np.array(res[0]["evals"]) - one.evals)) <= 1e-10 * one.params["aNorm"]
    assert abs(sum(r["evecs_norm2"] for r in res) - 4.0) < 1e-8


def test_rccl_svds_rows_split(built, tmp_path):
    world, res = _launch("svds", tmp_path)
    m, n, k = 4000 * world, 500 * world, 5
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    one = svds(m, n, (rp, ci, va), numSvals=k, eps=1e-10, methodStage1="GD_plusK", backend="hip", maxBlockSize=2)
    assert one.ret == 0
    assert np.max(np.abs(np.array(res[0]["evals"]) - one.svals)) <= 1e-10 * one.svals[0]
    assert np.all(np.array(res[0]["resNorms"]) <= 2e-10 * res[0]["aNorm"])
    assert abs(sum(r["evecs_norm2"] for r in res) - k) < 1e-8 and abs(sum(r["u_norm2"] for r in res) - k) < 1e-8
