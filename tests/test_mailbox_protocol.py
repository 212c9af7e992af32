"""The hand-off logic of the peer-to-peer reduction (primme_amd/csrc/hipk_internal.h: hipk_xr_exchange, hipk_xr_next_seq;
csrc/comm_ipc.hip) on a box without a GPU: tests/csrc/mailbox_protocol_sim.c restates it with C11 atomics, one thread per
rank, with ranks that run ahead of and behind each other.  Checked: every rank gets the rank-ordered sum (identical bits)
for thousands of back-to-back reductions through the TWO slot generations, nobody waits forever, and the sequence counter's
wrap-around at 2^32 keeps the generations alternating (with the naive rule `0 -> 1` the simulation deadlocks there: two
consecutive reductions would share a generation).  The device execution of the same protocol with 2, 4 and 8 processes is
tests/test_multirank_ipc_gpu.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mbsim") / "mailbox_protocol_sim")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-pthread",
                           os.path.join(HERE, "csrc", "mailbox_protocol_sim.c"), "-o", exe])
    return exe


@pytest.mark.parametrize("ranks,reductions,count,first", [(2, 2000, 5, 0), (4, 2000, 3, 0), (8, 1000, 2, 0), (16, 300, 1, 0),
                                                        (8, 300, 1, 4294967290), (3, 400, 4, 4294967293)])
def test_rank_ordered_sums_through_two_generations(sim, ranks, reductions, count, first):
    r = subprocess.run([sim, str(ranks), str(reductions), str(count), str(first), "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout


@pytest.fixture(scope="module")
def halosim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("halosim") / "halo_protocol_sim")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-pthread",
                           os.path.join(HERE, "csrc", "halo_protocol_sim.c"), "-o", exe])
    return exe


@pytest.mark.parametrize("ranks,exchanges,rows", [(2, 2000, 64), (4, 1500, 257), (8, 800, 33)])
def test_halo_landing_zones_two_generations(halosim, ranks, exchanges, rows):
    """The neighbour exchange (xr_halo_kernel): rows pushed into the neighbour's zone of generation seq & 1, one flag per
    exchange, the consumer reads while faster neighbours already push the next one.  (With ONE generation the same
    simulation reads a million overwritten words.)"""
    r = subprocess.run([halosim, str(ranks), str(exchanges), str(rows)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 bad words" in r.stdout


@pytest.fixture(scope="module")
def winsim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("winsim") / "window_protocol_sim")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-pthread",
                           os.path.join(HERE, "csrc", "window_protocol_sim.c"), "-o", exe])
    return exe


@pytest.mark.parametrize("ranks,ops,words", [(2, 1500, 100), (4, 1000, 64), (8, 500, 33)])
def test_bulk_window_one_barrier_two_generations(winsim, ranks, ops, words):
    """The bulk all-gather / reduce-scatter (push into every rank's window, ONE device-side barrier, local read): two window
    generations make the single barrier per operation enough."""
    r = subprocess.run([winsim, str(ranks), str(ops), str(words)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 bad words" in r.stdout
