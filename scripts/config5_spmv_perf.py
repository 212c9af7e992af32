"""SpMV timings of the BASELINE configs[4] operator pair (A: 8M x 2M, 5 nnz/row; A': 2M x 8M) in isolation.
usage: python scripts/config5_spmv_perf.py [rows cols]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from primme_amd import _ffi as F, problems
from primme_amd.svds_api import transpose_csr
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
lib = F.load_product()
ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
rp, ci, va = problems.svds_synthetic_csr(m, n)
rpT, ciT, vaT = transpose_csr(m, n, rp, ci, va)
dt = F.HIPK_F64
A = C.c_void_p(); At = C.c_void_p()
assert lib.hipk_csr_create_rect(ctx, dt, m, n, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(A)) == 0
assert lib.hipk_csr_create_rect(ctx, dt, n, m, rpT.ctypes.data_as(C.c_void_p), ciT.ctypes.data_as(C.c_void_p), vaT.ctypes.data_as(C.c_void_p), C.byref(At)) == 0
x = torch.randn(n, dtype=torch.float64, device="cuda"); u = torch.zeros(m, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def timeit(fn, nbytes, label, reps=20):
    for _ in range(3): fn()
    lib.hipk_sync(ctx)
    ms = C.c_float(); lib.hipk_timer_start(ctx)
    for _ in range(reps): fn()
    lib.hipk_timer_stop(ctx, C.byref(ms))
    us = 1e3 * ms.value / reps
    print(f"{label:40s} {us:9.1f} us  {nbytes / us / 1e3:7.0f} GB/s (algorithmic)")
nnz = len(va)
timeit(lambda: lib.hipk_csr_matvec(A, None, x.data_ptr(), n, u.data_ptr(), m, 1), nnz * 12 + (m + 1) * 4 + (m + n) * 8, f"u = A x   ({m} x {n}, kind {lib.hipk_csr_kind(A)})")
timeit(lambda: lib.hipk_csr_matvec(At, None, u.data_ptr(), m, y.data_ptr(), n, 1), nnz * 12 + (n + 1) * 4 + (m + n) * 8, f"y = A' u  (kind {lib.hipk_csr_kind(At)})")
