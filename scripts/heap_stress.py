"""Heap-corruption hunt (round 4): one component per process, in a loop, under MALLOC_CHECK_=3 so that glibc aborts at the
first free/realloc next to a damaged chunk.   python scripts/heap_stress.py comm|solve|pb|csr [iterations]"""
import ctypes as C, gc, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from primme_amd import _ffi as F, problems
from primme_amd.api import Operator, Session, eigsh
what = sys.argv[1]; it = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = F.load_product()

def churn():
    junk = [bytearray(np.random.randint(1, 4000)) for _ in range(2000)]
    lst = []
    for i in range(20000): lst.append(i)
    del junk, lst; gc.collect()

def mkcomm():
    buf = (C.c_char * 128)(); assert lib.primme_amd_comm_unique_id(buf) == 0
    comm = C.c_void_p(); assert lib.primme_amd_comm_create(C.byref(comm), bytes(buf.raw), 0, 1) == 0
    return comm

if what == "comm":
    lib.primme_amd_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    for i in range(it):
        comm = mkcomm()
        x = torch.arange(1000, dtype=torch.float64, device="cuda"); y = torch.zeros_like(x)
        assert lib.primme_amd_comm_allgather(comm, None, x.data_ptr(), y.data_ptr(), 8000) == 0
        torch.cuda.synchronize()
        lib.primme_amd_comm_destroy(comm); churn()
elif what == "solve":
    rp, ci, va, n = problems.laplacian_csr((40, 41)); op = Operator(n, csr=(rp, ci, va)); v0 = problems.start_vector(n)
    os.environ["PRIMME_AMD_FORCE_COMM"] = "1"
    for i in range(it):
        comm = mkcomm()
        r = eigsh(op, backend="hip", v0=v0, comm=comm, numEvals=4, eps=1e-9, aNorm=8.0, maxBlockSize=(2 if i % 2 else 0), method=("JDQMR" if i % 3 == 0 else "GD_plusK"))
        assert r.ret == 0
        lib.primme_amd_comm_destroy(comm); churn()
elif what == "nocomm":
    rp, ci, va, n = problems.laplacian_csr((40, 41)); op = Operator(n, csr=(rp, ci, va)); v0 = problems.start_vector(n)
    for i in range(it):
        r = eigsh(op, backend="hip", v0=v0, numEvals=4, eps=1e-9, aNorm=8.0, maxBlockSize=(2 if i % 2 else 0), method=("JDQMR" if i % 3 == 0 else "GD_plusK"))
        assert r.ret == 0; churn()
elif what == "pb":
    ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
    m, n = 300_000, 900_000
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.zeros(m, dtype=torch.float64, device="cuda")
    for i in range(it):
        A = C.c_void_p()
        assert lib.hipk_csr_create_rect(ctx, F.HIPK_F64, m, n, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(A)) == 0
        assert lib.hipk_csr_matvec(A, None, x.data_ptr(), n, y.data_ptr(), m, 1) == 0
        lib.hipk_sync(ctx); lib.hipk_csr_destroy(A); churn()
print(what, "done", it)
