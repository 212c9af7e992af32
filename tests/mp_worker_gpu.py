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

    res = dict(rank=rank, case=case)
    if case in ("halo", "halo_block"):
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
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.barrier()
    lib.primme_amd_comm_destroy(comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
