"""Worker of tests/test_multigpu_rccl.py: one process per GPU (rank r drives cuda:r), the library's own
RCCL communicator (primme_amd_comm: in-stream all-reduce of the <= 4 KB partials, neighbour halo with
grouped ncclSend/ncclRecv, grouped all-gather / reduce-scatter), rows partitioned as in the
reference's MPI example (examples/ex_eigs_mpi.c:100-123).  The 128-byte RCCL id travels over a gloo
process group (rendezvous on 127.0.0.1).  Every case writes its results to <out>.<rank> as JSON."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure


def split(n, world, rank):
    base, rem = divmod(n, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def comm_ops(lib, comm, rank, world, res):
    """Every collective of include/primme_amd_comm.h with known data (whatever transport the communicator has)."""
    import time
    import torch
    lib.primme_amd_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.primme_amd_comm_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_size_t, C.c_int64,
                                         C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.primme_amd_comm_allgather_cols.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_size_t, C.c_size_t, C.c_int]
    lib.primme_amd_comm_reduce_scatter_cols.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_size_t, C.c_int, C.c_int]
    lib.primme_amd_comm_allgather_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.primme_amd_comm_error.argtypes = [C.c_void_p]
    ok = True
    gen = lambda r, n, salt: np.random.default_rng(1000 * salt + r).standard_normal(n)
    # all-reduce: sizes around the granule slot (4096) and the 16-element blocks; rank-ordered sum, bit for bit
    digests = []
    for salt, cnt in enumerate([1, 2, 15, 16, 17, 300, 4096, 4097, 10000]):
        x = torch.from_numpy(gen(rank, cnt, salt)).cuda()
        assert lib.primme_amd_comm_allreduce(comm, None, x.data_ptr(), cnt) == 0
        torch.cuda.synchronize()
        want = np.zeros(cnt)
        for r in range(world):
            want = want + gen(r, cnt, salt)
        got = x.cpu().numpy()
        exact = np.array_equal(got, want)
        ok &= bool(exact or np.allclose(got, want, rtol=0, atol=1e-12))
        res.setdefault("allreduce_exact", []).append(bool(exact))
        digests.append(got.tobytes().hex()[:64] + str(float(got.sum())))
    res["digest"] = digests
    # neighbour halo, several columns, uneven counts, twice in a row (generations)
    nrows, ncols, ld = 1000 + 13 * rank, 3, 1100
    for it in range(3):
        xh = np.zeros((ncols, ld)); xh[:, :nrows] = (rank + 1) * 1000 + np.arange(nrows)[None, :] + 0.5 * np.arange(ncols)[:, None] + 0.25 * it
        x = torch.from_numpy(xh).cuda()
        lo_n, hi_n = 7, 5     # every rank needs 7 rows from below and 5 from above -> sends 5 down... (send_lo = what r-1 needs from above = 5)
        lo = torch.zeros(ncols * lo_n, dtype=torch.float64, device="cuda"); hi = torch.zeros(ncols * hi_n, dtype=torch.float64, device="cuda")
        assert lib.primme_amd_comm_halo(comm, None, x.data_ptr(), ld, nrows, ncols, 8, hi_n, lo_n, lo.data_ptr(), lo_n, hi.data_ptr(), hi_n) == 0
        torch.cuda.synchronize()
        if rank > 0:
            nr = 1000 + 13 * (rank - 1)
            want = (rank) * 1000 + np.arange(nr - lo_n, nr)[None, :] + 0.5 * np.arange(ncols)[:, None] + 0.25 * it
            ok &= bool(np.array_equal(lo.cpu().numpy().reshape(ncols, lo_n), want))
        if rank < world - 1:
            want = (rank + 2) * 1000 + np.arange(hi_n)[None, :] + 0.5 * np.arange(ncols)[:, None] + 0.25 * it
            ok &= bool(np.array_equal(hi.cpu().numpy().reshape(ncols, hi_n), want))
    # all-gather / reduce-scatter of column blocks (double and float), repeated (window generations)
    for it in range(3):
        per, nc = 257 + it, 2
        send = torch.from_numpy(np.stack([gen(rank, per, 50 + c + 10 * it) for c in range(nc)])).cuda()
        recv = torch.zeros((nc, per * world + 3), dtype=torch.float64, device="cuda")
        assert lib.primme_amd_comm_allgather_cols(comm, None, send.data_ptr(), per, recv.data_ptr(), per * world + 3, per * 8, 8, nc) == 0
        torch.cuda.synchronize()
        want = np.stack([np.concatenate([gen(r, per, 50 + c + 10 * it) for r in range(world)]) for c in range(nc)])
        ok &= bool(np.array_equal(recv.cpu().numpy()[:, :per * world], want))
        for dt, isd in ((torch.float64, 1), (torch.float32, 0)):
            full = torch.from_numpy(np.stack([gen(rank, per * world, 70 + c + 10 * it) for c in range(nc)])).to(dt).cuda()
            out = torch.zeros((nc, per), dtype=dt, device="cuda")
            assert lib.primme_amd_comm_reduce_scatter_cols(comm, None, full.data_ptr(), per * world, out.data_ptr(), per, per, isd, nc) == 0
            torch.cuda.synchronize()
            npdt = np.float64 if isd else np.float32
            want = np.zeros((nc, per), dtype=npdt)
            for r in range(world):
                want = want + np.stack([gen(r, per * world, 70 + c + 10 * it) for c in range(nc)]).astype(npdt)[:, rank * per:(rank + 1) * per]
            ok &= bool(np.allclose(out.cpu().numpy(), want, rtol=0, atol=1e-12 if isd else 1e-5))
    mine = (C.c_int64 * 40)(*[100 * rank + i for i in range(40)]); allv = (C.c_int64 * (40 * world))()
    assert lib.primme_amd_comm_allgather_i64(comm, mine, 40, allv) == 0
    ok &= list(allv) == [100 * r + i for r in range(world) for i in range(40)]
    # latency of one small reduction: 200 back-to-back 8-double all-reduces, host clock around launch..sync
    x = torch.ones(8, dtype=torch.float64, device="cuda")
    for _ in range(20):
        lib.primme_amd_comm_allreduce(comm, None, x.data_ptr(), 8)
    x.fill_(1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        lib.primme_amd_comm_allreduce(comm, None, x.data_ptr(), 8)
        torch.cuda.synchronize()
    res["allreduce8_sync_us"] = 1e6 * (time.perf_counter() - t0) / 200
    x.fill_(1.0 / 1024); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lib.primme_amd_comm_allreduce(comm, None, x.data_ptr(), 8)
    torch.cuda.synchronize()
    res["allreduce8_chain_us"] = 1e6 * (time.perf_counter() - t0) / 5
    ok &= bool(np.all(x.cpu().numpy() == (1.0 / 1024) * float(world) ** 5))
    ok &= lib.primme_amd_comm_error(comm) == 0
    # the library's own self-test (what bench.py --gpus N runs before its timed region)
    lib.primme_amd_comm_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    us = C.c_double(-1.0)
    res["selftest"] = lib.primme_amd_comm_selftest(comm, None, 200, C.byref(us))
    res["selftest_allreduce_us"] = us.value
    ok &= res["selftest"] == 0 and us.value > 0
    res.update(ret=0 if ok else 1, evals=[], resNorms=[], its=0, numGlobalSum=1, evecs_norm2=0.0, aNorm=0.0)
    return None


def run(rank, world, port, case, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank % max(torch.cuda.device_count(), 1))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from primme_amd import _ffi as F, problems
    from primme_amd.api import Operator, Session
    lib = F.load_product()
    if world == 1:
        os.environ["PRIMME_AMD_FORCE_COMM"] = "1"      # one GPU: still go through the communicator
    buf = (C.c_char * 128)()
    if rank == 0:
        assert lib.primme_amd_comm_unique_id(buf) == 0
    uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(uid, 0)
    comm = C.c_void_p()
    assert lib.primme_amd_comm_create(C.byref(comm), bytes(uid.numpy().tobytes()), rank, world) == 0

    lib.primme_amd_comm_transport.restype = C.c_char_p
    lib.primme_amd_comm_transport.argtypes = [C.c_void_p]
    res = dict(rank=rank, case=case, transport=lib.primme_amd_comm_transport(comm).decode())
    if case == "comm_ops":
        r = comm_ops(lib, comm, rank, world, res)
    elif case == "lap3d_small":
        # BASELINE configs[1] in small: 3-D 7-point Laplacian, 10 smallest, GD+k, block size 1
        dims = (40, 41, 42)
        n = int(np.prod(dims))
        row0, nloc = split(n, world, rank)
        rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc), comm=comm)
        r = s.solve(numEvals=10, eps=1e-8, aNorm=12.0, method="GD_plusK", numProcs=world, procID=rank,
                    v0=problems.start_vector(n, row0=row0, nrows=nloc))
        s.close()
    elif case == "hermitian_b4":
        # BASELINE configs[3] in small: complex Hermitian banded, block size 4, 6 largest, GD+k, rows split
        nloc = 4000
        n = nloc * world
        rp, ci, va = problems.hermitian_banded_csr(n, row0=rank * nloc, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=rank * nloc, nrows=nloc), comm=comm, dtype=np.complex128)
        r = s.solve(numEvals=6, target="largest", eps=1e-10, numProcs=world, procID=rank, iseed=(5, 1, 2, 3), maxBlockSize=4,
                    method="GD_plusK")
        s.close()
        r.evecs = np.concatenate([r.evecs.real, r.evecs.imag])
    elif case in ("halo", "halo_block"):
        # one 3-D Laplacian split by rows: neighbour halo exchange inside the ready-made operator
        dims = (24, 25, 26)
        n = int(np.prod(dims))
        row0, nloc = split(n, world, rank)
        rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc), comm=comm)
        kw = dict(numEvals=6, eps=1e-10, aNorm=12.0, numProcs=world, procID=rank)
        if case == "halo":
            r = s.solve(v0=problems.start_vector(n, row0=row0, nrows=nloc), **kw)
        else:
            v0 = np.random.default_rng(7).standard_normal((n, 4))[row0:row0 + nloc]
            r = s.solve(v0=v0, maxBlockSize=4, method="JDQMR", **kw)
        s.close()
    elif case in ("halo_mass", "halo_mass_jdqmr"):
        # generalised problem (round 6): the 3-D Laplacian and a tridiagonal mass matrix, both split by rows — two ready-made
        # operators with their own neighbour exchanges, B X / B (V h) / B d through primme_amd_mass_matvec on every rank
        dims = (24, 25, 26)
        n = int(np.prod(dims))
        row0, nloc = split(n, world, rank)
        rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
        brp, bci, bva = problems.mass_matrix_csr(n, row0=row0, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc), comm=comm, mass=Operator(n, csr=(brp, bci, bva), row0=row0, nrows=nloc))
        kw = dict(method="JDQMR", precond="jacobi", locking=1) if case.endswith("_jdqmr") else dict(method="GD_plusK")
        r = s.solve(numEvals=5, eps=1e-9, aNorm=12.0, numProcs=world, procID=rank, v0=problems.start_vector(n, row0=row0, nrows=nloc), **kw)
        s.close()
    elif case in ("config2_full", "lap2d_10m"):
        # the bench workloads under the bench's own row partition (bench.py --gpus N): BASELINE configs[1] at full
        # size, and the north-star 10 M-row 5-point Laplacian (2 pairs: the partition is what is under test)
        dims, nev, aN = ((125, 126, 127), 10, 12.0) if case == "config2_full" else ((3162, 3163), 2, 8.0)
        n = int(np.prod(dims))
        row0, nloc = split(n, world, rank)
        rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc), comm=comm)
        r = s.solve(numEvals=nev, eps=1e-8, aNorm=aN, method="GD_plusK", numProcs=world, procID=rank,
                    v0=problems.start_vector(n, row0=row0, nrows=nloc))
        s.close()
    elif case == "blockdiag":
        dims = (40, 41)
        rp, ci, va, n0 = problems.laplacian_csr(dims)
        rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 1, scale_fn=lambda t: 1.0 + 0.37 * t / max(world - 1, 1), row0_tile=rank)
        n = n0 * world
        s = Session(Operator(n, csr=(rpt, cit, vat), row0=rank * n0, nrows=n0), comm=comm)
        r = s.solve(numEvals=6, eps=1e-10, aNorm=8.0 * 1.37, v0=problems.start_vector(n, row0=rank * n0, nrows=n0),
                    numProcs=world, procID=rank)
        s.close()
    elif case == "allgather":
        # Laplacian plus a symmetric long-range coupling i <-> i + n/2: columns far outside the neighbours'
        # slabs, so the operator gathers the whole vector (one grouped exchange per block)
        dims = (64, 8 * world)
        rp, ci, va, n = problems.laplacian_csr(dims)
        import scipy.sparse as sp
        A = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
        h = n // 2
        for i in range(0, h, 5):
            A[i, i + h] = 0.25; A[i + h, i] = 0.25
        A = A.tocsr(); A.sort_indices()
        row0, nloc = split(n, world, rank)
        lrp = (A.indptr[row0:row0 + nloc + 1] - A.indptr[row0]).astype(np.int32)
        lci = A.indices[A.indptr[row0]:A.indptr[row0 + nloc]].astype(np.int32)
        lva = A.data[A.indptr[row0]:A.indptr[row0 + nloc]].astype(np.float64)
        s = Session(Operator(n, csr=(lrp, lci, lva), row0=row0, nrows=nloc), comm=comm)
        v0 = np.random.default_rng(3).standard_normal((n, 2))[row0:row0 + nloc]
        r = s.solve(numEvals=4, eps=1e-10, aNorm=8.5, v0=v0, maxBlockSize=2, numProcs=world, procID=rank)
        s.close()
    elif case == "hermitian":
        nloc = 3000
        n = nloc * world
        rp, ci, va = problems.hermitian_banded_csr(n, row0=rank * nloc, nrows=nloc)
        s = Session(Operator(n, csr=(rp, ci, va), row0=rank * nloc, nrows=nloc), comm=comm, dtype=np.complex128)
        r = s.solve(numEvals=4, target="largest", eps=1e-10, numProcs=world, procID=rank, iseed=(5, 1, 2, 3), maxBlockSize=2)
        s.close()
        r.evecs = np.concatenate([r.evecs.real, r.evecs.imag])
    elif case == "svds":
        # A split by rows, n-vectors in equal slabs: grouped all-gather / reduce-scatter per block
        m, n, k = 4000 * world, 500 * world, 5
        rp, ci, va = problems.svds_synthetic_csr(m, n)
        mloc, nloc = m // world, n // world
        r0 = rank * mloc
        lrp = (rp[r0:r0 + mloc + 1] - rp[r0]).astype(np.int32)
        lci = np.ascontiguousarray(ci[rp[r0]:rp[r0 + mloc]]); lva = np.ascontiguousarray(va[rp[r0]:rp[r0 + mloc]])
        ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
        oph = C.c_void_p()
        lib.primme_amd_svds_operator_create_dist.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert lib.primme_amd_svds_operator_create_dist(C.byref(oph), ctx, F.HIPK_F64, mloc, n, nloc, lrp.ctypes.data, lci.ctypes.data,
                                                        lva.ctypes.data, comm) == 0
        ps = F.PrimmeSvdsParams()
        lib.primme_svds_initialize(C.byref(ps))
        ps.m, ps.n, ps.numSvals, ps.eps, ps.printLevel, ps.outputFile = m, n, k, 1e-10, 0, None
        ps.numProcs, ps.procID, ps.mLocal, ps.nLocal = world, rank, mloc, nloc
        ps.matrix = oph
        ps.matrixMatvec = C.cast(lib.primme_amd_svds_matvec, C.c_void_p)
        ps.commInfo = comm
        ps.globalSumReal = C.cast(lib.primme_amd_svds_global_sum, C.c_void_p)
        ps.maxBlockSize = 2
        lib.primme_svds_set_method(F.SVDS_METHODS["normalequations"], F.METHODS["GD_plusK"], 0, C.byref(ps))
        svals, rn = np.zeros(k), np.zeros(k)
        sv = torch.zeros((mloc + nloc) * k, dtype=torch.float64, device="cuda")
        ret = lib.hip_dprimme_svds(svals.ctypes.data_as(C.c_void_p), C.c_void_p(sv.data_ptr()), rn.ctypes.data_as(C.c_void_p), C.byref(ps))
        torch.cuda.synchronize()
        svh = sv.cpu().numpy()
        U, V = svh[:mloc * k].reshape(k, mloc), svh[mloc * k:].reshape(k, nloc)
        res.update(ret=ret, evals=svals.tolist(), resNorms=rn.tolist(), its=int(ps.stats.numOuterIterations),
                   numGlobalSum=int(ps.stats.numGlobalSum), evecs_norm2=float(np.sum(V ** 2)), u_norm2=float(np.sum(U ** 2)),
                   aNorm=float(ps.aNorm))
        lib.primme_amd_svds_operator_destroy(oph); lib.hipk_ctx_destroy(ctx)
        r = None
    else:
        raise ValueError(case)
    if r is not None:
        res.update(ret=r.ret, evals=r.evals.tolist(), resNorms=r.resNorms.tolist(), its=r.stats["numOuterIterations"],
                   matvecs=r.stats["numMatvecs"], numGlobalSum=r.stats["numGlobalSum"], evecs_norm2=float(np.sum(np.abs(r.evecs) ** 2)),
                   aNorm=r.params["aNorm"])
        pre = (C.c_long * 2)()
        lib.primme_amd_prelaunch_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.primme_amd_prelaunch_stats(C.cast(pre, C.POINTER(C.c_long)), C.cast(C.byref(pre, C.sizeof(C.c_long)), C.POINTER(C.c_long)))
        res.update(ahead=int(pre[0]), adopted=int(pre[1]))      # iterations enqueued before the host had seen the previous one / adopted
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.barrier()
    lib.primme_amd_comm_destroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
