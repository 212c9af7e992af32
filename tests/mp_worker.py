"""Worker of the world_size-2 gloo tests: each rank owns a slab of rows (reference
examples/ex_eigs_mpi.c:100-123), reductions go through a user globalSumReal callback
(dist.all_reduce on host buffers, the reference's contract), the matvec exchanges halos with
dist.send/recv.  Runs the product's host solver over the plain-C kernel layer (no GPU)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure


def run(rank, world, port, case, out_path):
    import ctypes as C
    import torch
    extra = {}
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from primme_amd import problems
    from checkers import Operator
    from checkers import Session
    from primme_amd import _ffi as F
    import checkers

    def global_sum(a):
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t)
        return t.numpy()

    if case == "blockdiag":
        # two independent, differently scaled Laplacians: no matvec communication at all
        dims = (15, 16)
        rp, ci, va, n0 = problems.laplacian_csr(dims)
        rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 1, scale_fn=lambda t: 1.0 + 0.37 * t, row0_tile=rank)
        n = n0 * world
        op = Operator(n, csr=(rpt, cit, vat), row0=rank * n0, nrows=n0)
        v0 = problems.start_vector(n, row0=rank * n0, nrows=n0)
        s = Session(op, backend="hostcheck")
        r = s.solve(numEvals=6, eps=1e-10, aNorm=8.0 * 1.37, v0=v0, numProcs=world, procID=rank, global_sum=global_sum)
        s.close()
    elif case.startswith("blockdiag_mass"):
        # generalised problem A x = lambda B x with the rows over two ranks (round 6): A and the mass matrix B block-diagonal,
        # one tile per rank, so neither callback communicates; every inner product of the tracked-Gram path, of the
        # residuals W h - theta B(V h) and (…_jdqmr) of the inner solver's projectors on B Q / B x goes through globalSumReal
        dims = (15, 16)
        rp, ci, va, n0 = problems.laplacian_csr(dims)
        rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 1, scale_fn=lambda t: 1.0 + 0.37 * t, row0_tile=rank)
        brp, bci, bva = problems.mass_matrix_csr(n0)
        brpt, bcit, bvat = problems.tile_block_diagonal(brp, bci, bva, 1, scale_fn=lambda t: 1.0 + 0.11 * t, row0_tile=rank)
        n = n0 * world
        op = Operator(n, csr=(rpt, cit, vat), row0=rank * n0, nrows=n0)
        bop = Operator(n, csr=(brpt, bcit, bvat), row0=rank * n0, nrows=n0)
        v0 = problems.start_vector(n, row0=rank * n0, nrows=n0)
        s = Session(op, backend="hostcheck", mass=bop)
        kw = dict(method="JDQMR", precond="jacobi", locking=1) if case.endswith("_jdqmr") else dict(method="GD_plusK")
        r = s.solve(numEvals=5, eps=1e-9, aNorm=8.0 * 1.37, v0=v0, numProcs=world, procID=rank, global_sum=global_sum, **kw)
        s.close()
    elif case.startswith("devcomm"):
        xr = case.endswith("_xr")
        case = case[:-3] if xr else case
        # The path the GPUs take: the library's own operator and communicator (reductions inside the stream of
        # launches, |t|^2 and t'At in one all-reduce, fused / speculative restart on reduced overlaps).  The
        # all-reduce of the stand-in communicator (oracle/hostcheck_glue.c) is gloo; block-diagonal matrix, so
        # the operator itself needs no exchange.  devcomm_lock: locking (10 pairs), devcomm_soft: 4 pairs.
        dims = (15, 16)
        rp, ci, va, n0 = problems.laplacian_csr(dims)
        rpt, cit, vat = problems.tile_block_diagonal(rp, ci, va, 1, scale_fn=lambda t: 1.0 + 0.37 * t, row0_tile=rank)
        n = n0 * world
        if case == "devcomm_halo":
            # ONE 2-D Laplacian split by rows: the slabs reference each other's boundary rows, so the operator exchanges
            # halo data and the one-launch tail (scale + A t + t'At) runs with halo buffers, as on several GPUs
            dims = (24, 22)
            n = dims[0] * dims[1]
            n0 = n // world
            rpt, cit, vat, _ = problems.laplacian_csr(dims, row0=rank * n0, nrows=n0)
        lib = checkers.load_hostcheck()
        AR = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int)

        def allreduce(buf, count):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t)
            a[:] = t.numpy()
        arcb = AR(allreduce)
        comm = C.c_void_p()
        lib.primme_amd_hostcheck_comm_create.argtypes = [C.POINTER(C.c_void_p), AR, C.c_int, C.c_int]
        assert lib.primme_amd_hostcheck_comm_create(C.byref(comm), arcb, rank, world) == 0
        if xr:
            # stand-in of the peer-to-peer transport's fused second stage: armed reductions are summed over the ranks inside the
            # "launch" that forms them (oracle/hostcheck_glue.c), the host logic is the one of ranks on the mailboxes
            lib.primme_amd_hostcheck_comm_set_xr.argtypes = [C.c_void_p, C.c_int]
            assert lib.primme_amd_hostcheck_comm_set_xr(comm, 1) == 0
        op = Operator(n, csr=(rpt, cit, vat), row0=rank * n0, nrows=n0)
        v0 = problems.start_vector(n, row0=rank * n0, nrows=n0)
        s = Session(op, comm=comm, backend="hostcheck")
        nev = 10 if case == "devcomm_lock" else (6 if case in ("devcomm_halo", "devcomm_jdqmr", "devcomm_jdqmr3") else 4)
        counts = (C.c_long * 8)()
        lib.hipk_cpu_counts(counts, 1)
        if case in ("devcomm_jdqmr", "devcomm_jdqmr3"):
            # block JDQMR with the library's Jacobi preconditioner, rows over two ranks: the inner step with ONE synchronisation
            # (scalar recurrences evaluated next to the launches, from all-reduced sums) — or, "3", the three-wait sequence
            if case == "devcomm_jdqmr3":
                os.environ["PRIMME_AMD_QMR_THREE_WAITS"] = "1"
            v0b = np.random.default_rng(7).standard_normal((n, 4))[rank * n0:(rank + 1) * n0]
            r = s.solve(numEvals=nev, eps=1e-9, aNorm=8.0 * 1.37, v0=v0b, numProcs=world, procID=rank, method="JDQMR", maxBlockSize=4,
                        precond=("jacobi", 0.0))
        else:
            r = s.solve(numEvals=nev, eps=1e-10, aNorm=8.0 * (1.0 if case == "devcomm_halo" else 1.37), v0=v0, numProcs=world, procID=rank)
        lib.hipk_cpu_counts(counts, 1)
        lib.primme_amd_hostcheck_comm_calls.restype = C.c_long
        lib.primme_amd_hostcheck_comm_calls.argtypes = [C.c_void_p]
        lib.primme_amd_hostcheck_comm_xr_calls.restype = C.c_long
        lib.primme_amd_hostcheck_comm_xr_calls.argtypes = [C.c_void_p]
        pre = (C.c_long * 2)()
        lib.primme_amd_prelaunch_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.primme_amd_prelaunch_stats(C.cast(pre, C.POINTER(C.c_long)), C.cast(C.byref(pre, C.sizeof(C.c_long)), C.POINTER(C.c_long)))
        qs = (C.c_long * 2)()
        lib.primme_amd_qmr_step_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.primme_amd_qmr_step_stats(C.cast(qs, C.POINTER(C.c_long)), C.cast(C.byref(qs, C.sizeof(C.c_long)), C.POINTER(C.c_long)))
        extra = dict(qmr_steps=int(qs[0]), qmr_steps_one_wait=int(qs[1]),
                     allreduces=int(lib.primme_amd_hostcheck_comm_calls(comm)), fused_tail=int(counts[5]), ritz_cgs=int(counts[3]),
                     ritz_ov=int(counts[6]), dots=int(counts[0]), locking=int(r.params["locking"]),
                     fused_exchanges=int(lib.primme_amd_hostcheck_comm_xr_calls(comm)), ahead=int(pre[0]), adopted=int(pre[1]))
        s.close()
    elif case == "halo":
        # one 2-D Laplacian split by rows; the callback matvec exchanges one grid line with the neighbour
        dims = (20, 22)
        n = dims[0] * dims[1]
        base = n // world
        row0, nloc = rank * base, base if rank < world - 1 else n - rank * base
        rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
        lo = row0 - max(0, min(ci.min(), row0)) if row0 > 0 else 0
        hi = max(0, int(ci.max()) - (row0 + nloc) + 1)
        lib = checkers.load_hostcheck()
        s = Session(Operator(n, csr=(rp, ci, va), row0=row0, nrows=nloc), backend="hostcheck")
        A = [h for k, h in s.handles if k == "csr"][0]
        assert lib.hipk_csr_halo_lo(A) == lo and lib.hipk_csr_halo_hi(A) == hi

        def matvec(x, ldx, y, ldy, bs, pp, ierr):
            nb, lx, ly = bs[0], ldx[0], ldy[0]
            X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(nb, lx))
            Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(nb, ly))
            for c in range(nb):
                xl = np.ascontiguousarray(X[c, :nloc])
                lo_buf, hi_buf = np.zeros(max(lo, 1)), np.zeros(max(hi, 1))
                reqs = []
                if rank > 0:
                    reqs.append(dist.isend(torch.from_numpy(xl[:hi_of[rank - 1]].copy()), rank - 1))
                    tlo = torch.zeros(lo, dtype=torch.float64); reqs.append(dist.irecv(tlo, rank - 1))
                if rank < world - 1:
                    reqs.append(dist.isend(torch.from_numpy(xl[nloc - lo_of[rank + 1]:].copy()), rank + 1))
                    thi = torch.zeros(hi, dtype=torch.float64); reqs.append(dist.irecv(thi, rank + 1))
                for q in reqs: q.wait()
                if rank > 0: lo_buf[:lo] = tlo.numpy()
                if rank < world - 1: hi_buf[:hi] = thi.numpy()
                lib.hipk_csr_set_halo(A, lo_buf.ctypes.data_as(C.c_void_p), hi_buf.ctypes.data_as(C.c_void_p))
                yl = np.zeros(nloc)
                lib.hipk_csr_matvec(A, None, xl.ctypes.data_as(C.c_void_p), nloc, yl.ctypes.data_as(C.c_void_p), nloc, 1)
                Y[c, :nloc] = yl
            ierr[0] = 0

        # neighbours' halo sizes
        mine = torch.tensor([lo, hi], dtype=torch.int64)
        allh = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allh, mine)
        lo_of = [int(t[0]) for t in allh]
        hi_of = [int(t[1]) for t in allh]
        cb = F.BLOCK_OP(matvec)
        v0 = problems.start_vector(n, row0=row0, nrows=nloc)
        r = s.solve(numEvals=5, eps=1e-10, aNorm=8.0, v0=v0, numProcs=world, procID=rank, global_sum=global_sum,
                    user_matvec=cb)
        s.close()
    elif case == "zdevcomm":
        # BASELINE configs[3] in small on the path the GPUs take: complex panels, the library's own complex CSR operator and
        # communicator (all-reduce of the (re, im) partial sums inside the stream of launches), block size 4
        nloc = 150
        n = nloc * world
        rp, ci, va = problems.hermitian_banded_csr(nloc)
        va = va * (1.0 + 0.21 * rank)
        lib = checkers.load_hostcheck()
        AR = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int)

        def allreduce(buf, count):
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a.copy())
            dist.all_reduce(t)
            a[:] = t.numpy()
        arcb = AR(allreduce)
        comm = C.c_void_p()
        lib.primme_amd_hostcheck_comm_create.argtypes = [C.POINTER(C.c_void_p), AR, C.c_int, C.c_int]
        assert lib.primme_amd_hostcheck_comm_create(C.byref(comm), arcb, rank, world) == 0
        op = Operator(n, csr=(rp, (ci + rank * nloc).astype(np.int32), va), row0=rank * nloc, nrows=nloc)
        s = Session(op, comm=comm, backend="hostcheck", dtype=np.complex128)
        r = s.solve(numEvals=6, target="largest", eps=1e-9, maxBlockSize=4, maxBasisSize=20, minRestartSize=8, numProcs=world, procID=rank,
                    iseed=(2, 3, 5, 7))
        lib.primme_amd_hostcheck_comm_calls.restype = C.c_long
        lib.primme_amd_hostcheck_comm_calls.argtypes = [C.c_void_p]
        zextra = dict(allreduces=int(lib.primme_amd_hostcheck_comm_calls(comm)))
        s.close()
        r.evecs = np.concatenate([r.evecs.real, r.evecs.imag])
    elif case.startswith("hermitian"):
        # complex Hermitian band matrix (BASELINE configs[3] in small) cut into independent diagonal
        # blocks, one per rank: hip_zprimme's native complex solve (complex panels: every inner product is a
        # (re, im) pair in the reduction buffers) reduces through the user's globalSumReal.  Variants: the block
        # iteration of configs[3], JDQMR, harmonic and refined extraction for an interior target
        nloc = 150
        n = nloc * world
        rp, ci, va = problems.hermitian_banded_csr(nloc)
        va = va * (1.0 + 0.21 * rank)
        op = Operator(n, csr=(rp, (ci + rank * nloc).astype(np.int32), va), row0=rank * nloc, nrows=nloc)
        s = Session(op, backend="hostcheck", dtype=np.complex128)
        extra = {"hermitian": {}, "hermitian_blk4": dict(maxBlockSize=4, maxBasisSize=20, minRestartSize=8),
                 "hermitian_jdqmr": dict(method="JDQMR"),
                 "hermitian_harmonic": dict(target="closest_abs", targetShifts=[2.5], projection="harmonic", eps=1e-8),
                 "hermitian_refined": dict(target="closest_abs", targetShifts=[2.5], projection="refined", eps=1e-8)}[case]
        kw = dict(dict(numEvals=4, target="largest", eps=1e-10), **extra)
        r = s.solve(numProcs=world, procID=rank, global_sum=global_sum, iseed=(5 + rank, 1, 2, 3), **kw)
        s.close()
        r.evecs = np.concatenate([r.evecs.real, r.evecs.imag])
    elif case in ("svds", "svds_z"):
        # A (m x n) split by rows, n-vectors split in equal slabs: y = A x needs an all-gather of x,
        # y = A' x a reduce-scatter of the local products (BASELINE configs[4] in small).  svds_z: the same pattern with complex
        # phases through hip_zprimme_svds — the native complex front end (round 5) on two ranks
        from primme_amd.svds_api import transpose_csr
        cz = case == "svds_z"
        m, n, k = 600, 200, 4
        rp, ci, va = problems.svds_synthetic_csr(m, n)
        if cz:
            va = va.astype(np.complex128) * np.exp(1j * np.random.default_rng(5).uniform(0, 2 * np.pi, size=len(va)))
        mloc, nloc = m // world, n // world
        r0 = rank * mloc
        lrp = (rp[r0:r0 + mloc + 1] - rp[r0]).astype(np.int32)
        lci, lva = ci[rp[r0]:rp[r0 + mloc]], va[rp[r0]:rp[r0 + mloc]]
        if cz:
            import scipy.sparse as sp
            Al = sp.csr_matrix((lva, lci, lrp), shape=(mloc, n))
            AlH = Al.conj().T.tocsr()
        else:
            trp, tci, tva = transpose_csr(mloc, n, lrp, lci, lva)
        lib = checkers.load_hostcheck()
        sdt = np.complex128 if cz else np.float64
        tdt = torch.complex128 if cz else torch.float64

        def mv(x, ldx, y, ldy, bs, tr, pp, ierr):
            nb, lx, ly = bs[0], ldx[0], ldy[0]
            f = 2 if cz else 1                                 # leading dimensions count (complex) elements
            X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(nb, f * lx)).view(sdt)
            Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(nb, f * ly)).view(sdt)
            for c in range(nb):
                if not tr[0]:
                    parts = [torch.zeros(nloc, dtype=tdt) for _ in range(world)]
                    dist.all_gather(parts, torch.from_numpy(X[c, :nloc].copy()))
                    xf = torch.cat(parts).numpy()
                    Y[c, :mloc] = (Al @ xf) if cz else problems.csr_matvec_numpy(lrp, lci, lva, xf.reshape(-1, 1)).ravel()
                else:
                    zz = (AlH @ X[c, :mloc]) if cz else problems.csr_matvec_numpy(trp, tci, tva, X[c, :mloc].reshape(-1, 1)).ravel()
                    z = torch.from_numpy(np.ascontiguousarray(zz).copy())
                    dist.all_reduce(z)
                    Y[c, :nloc] = z.numpy()[rank * nloc:(rank + 1) * nloc]
            ierr[0] = 0

        GS = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(F.PrimmeSvdsParams), C.POINTER(C.c_int))

        def gs(send, recv, count, pp, ierr):
            a = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(count[0],))
            out = global_sum(a)
            np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(count[0],))[:] = out
            ierr[0] = 0
        cb, gcb = F.SVDS_BLOCK_OP(mv), GS(gs)
        ps = F.PrimmeSvdsParams()
        lib.primme_svds_initialize(C.byref(ps))
        ps.m, ps.n, ps.numSvals, ps.eps, ps.printLevel, ps.outputFile = m, n, k, 1e-10, 0, None
        ps.numProcs, ps.procID, ps.mLocal, ps.nLocal = world, rank, mloc, nloc
        ps.matrixMatvec = C.cast(cb, C.c_void_p)
        ps.globalSumReal = C.cast(gcb, C.c_void_p)
        lib.primme_svds_set_method(F.SVDS_METHODS["normalequations"], F.METHODS["GD_plusK"], 0, C.byref(ps))
        svals, rn = np.zeros(k), np.zeros(k)
        sv = np.zeros((mloc + nloc) * k, dtype=sdt)
        ret = (lib.hip_zprimme_svds if cz else lib.hip_dprimme_svds)(svals.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p),
                                                                     rn.ctypes.data_as(C.c_void_p), C.byref(ps))
        U = sv[:mloc * k].reshape(k, mloc)
        V = sv[mloc * k:].reshape(k, nloc)
        res = dict(rank=rank, ret=ret, evals=svals.tolist(), resNorms=rn.tolist(), its=int(ps.stats.numOuterIterations),
                   numGlobalSum=int(ps.stats.numGlobalSum), evecs_norm2=float(np.sum(np.abs(V) ** 2)), u_norm2=float(np.sum(np.abs(U) ** 2)),
                   aNorm=float(ps.aNorm), matvecs=int(ps.stats.numMatvecs))
        json.dump(res, open(f"{out_path}.{rank}", "w"))
        dist.barrier()
        dist.destroy_process_group()
        return
    else:
        raise ValueError(case)
    res = dict(rank=rank, ret=r.ret, evals=r.evals.tolist(), resNorms=r.resNorms.tolist(), its=r.stats["numOuterIterations"],
               numGlobalSum=r.stats["numGlobalSum"], evecs_norm2=float(np.sum(r.evecs ** 2)), restarts=r.stats["numRestarts"],
               matvecs=r.stats["numMatvecs"])
    if case.startswith("devcomm"):
        res.update(extra)
    if case == "zdevcomm":
        res.update(zextra)
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
