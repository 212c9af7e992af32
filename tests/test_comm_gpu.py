"""The RCCL communicator on the one GPU a test box has: a ONE-rank communicator exercises every call
the multi-rank path makes (ncclCommInitRank, in-stream ncclAllReduce of the device partials,
all-gather, grouped send/recv halo with no neighbours) and must not change any result.
Cross-rank arithmetic itself is covered by the world-size-2 gloo tests (tests/test_multirank_gloo.py);
two ranks on one device are rejected by RCCL ("Duplicate GPU detected")."""
import ctypes as C
import os

import numpy as np
import pytest

from primme_amd import _ffi as F
from primme_amd import problems
from checkers import eigsh, Operator

pytestmark = pytest.mark.gpu


def _comm(lib):
    buf = (C.c_char * 128)()
    assert lib.primme_amd_comm_unique_id(buf) == 0
    comm = C.c_void_p()
    assert lib.primme_amd_comm_create(C.byref(comm), bytes(buf.raw), 0, 1) == 0
    return comm


def test_one_rank_communicator_calls(built):
    import torch
    lib = F.load_product()
    comm = _comm(lib)
    lib.primme_amd_comm_rank.argtypes = [C.c_void_p]; lib.primme_amd_comm_size.argtypes = [C.c_void_p]
    assert lib.primme_amd_comm_rank(comm) == 0 and lib.primme_amd_comm_size(comm) == 1
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    y = torch.zeros_like(x)
    lib.primme_amd_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    assert lib.primme_amd_comm_allgather(comm, None, x.data_ptr(), y.data_ptr(), 8000) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    mine = (C.c_int64 * 2)(7, 9); allv = (C.c_int64 * 2)()
    lib.primme_amd_comm_allgather_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert lib.primme_amd_comm_allgather_i64(comm, mine, 2, allv) == 0 and list(allv) == [7, 9]
    lib.primme_amd_comm_destroy(comm)


@pytest.mark.parametrize("kw", [dict(numEvals=4, eps=1e-9, aNorm=8.0), dict(numEvals=3, eps=1e-9, aNorm=8.0, method="JDQMR"),
                                dict(numEvals=4, eps=1e-9, aNorm=8.0, maxBlockSize=2)])
def test_solver_through_rccl_reductions(built, kw):
    """Same solve with and without the device-communicator reduction path (forced on one rank)."""
    lib = F.load_product()
    rp, ci, va, n = problems.laplacian_csr((40, 41))
    op = Operator(n, csr=(rp, ci, va))
    v0 = problems.start_vector(n)
    a = eigsh(op, backend="hip", v0=v0, **kw)
    comm = _comm(lib)
    os.environ["PRIMME_AMD_FORCE_COMM"] = "1"
    try:
        b = eigsh(op, backend="hip", v0=v0, comm=comm, **kw)
    finally:
        del os.environ["PRIMME_AMD_FORCE_COMM"]
    assert a.ret == 0 and b.ret == 0
    assert b.stats["numGlobalSum"] > 0 and a.stats["numGlobalSum"] == 0
    # same arithmetic for the reductions (a one-rank all-reduce is the identity); the block-size-1 path applies
    # the operator to the un-normalised vector on this path (|t|^2 and t'At share one all-reduce), so the
    # new W column differs in the last bit: same history up to that
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-13 * 8.0 and np.max(np.abs(a.resNorms - b.resNorms)) <= 1e-10 * 8.0
    for k in ("numOuterIterations", "numMatvecs", "numRestarts"):
        assert abs(a.stats[k] - b.stats[k]) <= max(1, 0.02 * a.stats[k]), k
    lib.primme_amd_comm_destroy(comm)


@pytest.mark.parametrize("transport", ["auto", "rccl"])
def test_communicator_selftest_one_rank(built, transport, monkeypatch):
    """primme_amd_comm_selftest on the one GPU of the box (one rank: mailboxes under `auto`, RCCL when asked for): every
    collective against known data, and a latency figure"""
    lib = F.load_product()
    monkeypatch.setenv("PRIMME_AMD_COMM", transport)
    lib.primme_amd_comm_transport.restype = C.c_char_p
    lib.primme_amd_comm_transport.argtypes = [C.c_void_p]
    lib.primme_amd_comm_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    buf = (C.c_char * 128)()
    assert lib.primme_amd_comm_unique_id_for(buf, 1, 0) == 0
    comm = C.c_void_p()
    assert lib.primme_amd_comm_create(C.byref(comm), bytes(buf.raw), 0, 1) == 0
    assert lib.primme_amd_comm_transport(comm).decode() == ("rccl" if transport == "rccl" else "ipc")
    us = C.c_double(-1.0)
    assert lib.primme_amd_comm_selftest(comm, None, 100, C.byref(us)) == 0
    assert 0.0 < us.value < 1e4
    lib.primme_amd_comm_destroy(comm)


def test_mailboxes_that_do_not_come_up_fall_back_to_rccl(built):
    """`auto`: when the peer-to-peer mailboxes fail their self-test on any rank, every rank learns of it through the
    rendez-vous and the communicator comes up on RCCL instead (its id travels through the same rendez-vous)."""
    import torch
    lib = F.load_product()
    lib.primme_amd_comm_transport.restype = C.c_char_p
    lib.primme_amd_comm_transport.argtypes = [C.c_void_p]
    lib.primme_amd_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    ok = _comm(lib)
    assert lib.primme_amd_comm_transport(ok).decode() == "ipc"          # one rank: nothing for RCCL to add
    lib.primme_amd_comm_destroy(ok)
    os.environ["PRIMME_AMD_IPC_FAIL_SELFTEST"] = "1"
    try:
        comm = _comm(lib)
    finally:
        del os.environ["PRIMME_AMD_IPC_FAIL_SELFTEST"]
    assert lib.primme_amd_comm_transport(comm).decode() == "rccl"
    x = torch.arange(100, dtype=torch.float64, device="cuda")
    assert lib.primme_amd_comm_allreduce(comm, None, x.data_ptr(), 100) == 0
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(100, dtype=torch.float64))
    mine = (C.c_int64 * 2)(7, 9); allv = (C.c_int64 * 2)()
    lib.primme_amd_comm_allgather_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert lib.primme_amd_comm_allgather_i64(comm, mine, 2, allv) == 0 and list(allv) == [7, 9]
    lib.primme_amd_comm_destroy(comm)
