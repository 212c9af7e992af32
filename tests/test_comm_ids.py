"""Communicator ids and the fail-fast rules of primme_amd_comm_create (include/primme_amd_comm.h), no GPU needed: which
transport `auto` hands out for which job, that a mailbox id with more ranks than the mailboxes serve is refused on every
rank before anybody waits, that a rank which cannot see the rendez-vous segment (another node / IPC namespace) leaves at
once with advice, and that the ranks which can see it give up after the attach time limit, not after five minutes."""
import ctypes as C
import os
import time

import pytest

from primme_amd import _ffi as F


def _seg(buf):
    return "/dev/shm" + buf.raw[8:72].split(b"\0")[0].decode()


@pytest.fixture
def lib(built, monkeypatch):
    monkeypatch.delenv("PRIMME_AMD_COMM", raising=False)
    return F.load_product()


def test_auto_hands_out_what_can_serve_the_job(lib, monkeypatch):
    """(the decision is checked through primme_amd_comm_id_kind_for: making an ncclUniqueId starts RCCL's bootstrap threads, which a
    box without a GPU does not need — tests/test_comm_gpu.py brings the RCCL transport up on the device)"""
    kind = lib.primme_amd_comm_id_kind_for
    assert kind(8, 0) == 1 and kind(16, 0) == 1 and kind(1, 0) == 1      # one node, <= 16 ranks: mailboxes
    assert kind(17, 0) == 0 and kind(8, 1) == 0                           # more ranks / several nodes: RCCL, with default settings
    monkeypatch.setenv("PRIMME_AMD_COMM", "rccl")
    assert kind(2, 0) == 0
    monkeypatch.setenv("PRIMME_AMD_COMM", "ipc")
    assert kind(17, 0) == -43 and kind(4, 1) == -43 and kind(4, 0) == 1   # asked for explicitly and cannot be served: refused
    buf = (C.c_char * 128)()
    assert lib.primme_amd_comm_unique_id_for(buf, 17, 0) == -43
    assert lib.primme_amd_comm_unique_id_for(buf, 4, 0) == 0 and buf.raw[:6] == b"PAIPC1"
    os.unlink(_seg(buf))
    monkeypatch.delenv("PRIMME_AMD_COMM")
    assert lib.primme_amd_comm_unique_id_for(buf, 8, 0) == 0 and buf.raw[:6] == b"PAIPC1"
    os.unlink(_seg(buf))


def test_mailbox_id_with_too_many_ranks_fails_on_every_rank_without_waiting(lib):
    buf = (C.c_char * 128)()
    assert lib.primme_amd_comm_unique_id(buf) == 0 and buf.raw[:6] == b"PAIPC1"
    try:
        for rank in (0, 5, 19):
            comm = C.c_void_p()
            t0 = time.time()
            assert lib.primme_amd_comm_create(C.byref(comm), bytes(buf.raw), rank, 20) == -43
            assert time.time() - t0 < 2.0
    finally:
        os.unlink(_seg(buf))


def test_rank_that_cannot_see_the_segment_leaves_at_once_and_the_others_after_the_attach_limit(lib, monkeypatch):
    buf = (C.c_char * 128)()
    assert lib.primme_amd_comm_unique_id(buf) == 0
    seg = _seg(buf)
    raw = bytes(buf.raw)
    # "another node": the same id, but the segment is not there
    other = raw[:8] + (raw[8:72].split(b"\0")[0] + b"_elsewhere").ljust(64, b"\0") + raw[72:]
    comm = C.c_void_p()
    t0 = time.time()
    assert lib.primme_amd_comm_create(C.byref(comm), other, 1, 2) == -43
    assert time.time() - t0 < 2.0
    # the rank that does see it waits for the attach limit only
    monkeypatch.setenv("PRIMME_AMD_IPC_ATTACH_TIMEOUT_S", "1")
    t0 = time.time()
    assert lib.primme_amd_comm_create(C.byref(comm), raw, 0, 2) == -43
    assert 0.9 < time.time() - t0 < 10.0
    assert not os.path.exists(seg)          # rank 0 removes the name on its way out
