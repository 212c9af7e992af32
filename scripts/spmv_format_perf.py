"""One-column SpMV of the two Laplacian workloads in isolation, in both forms that can serve them: the CSR row-tile
kernel (csr_stream_kernel) and the row-pattern form (pat_kernel, csrc/hipk_sparse_pat.hip); plain product and the fused
tail of the block-size-1 iteration (scale + A t + t'At).  Prints microseconds per launch, GB/s on the plain-CSR
algorithmic bytes and on the bytes the form really moves, and checks that y is bit-identical between the forms.
usage: python scripts/spmv_format_perf.py [reps]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from primme_amd import _ffi as F, problems

lib = F.load_product()
lib.hipk_set_spmv_format.argtypes = [C.c_int]
lib.hipk_csr_format.argtypes = [C.c_void_p]; lib.hipk_csr_npatterns.argtypes = [C.c_void_p]
lib.hipk_csr_product_bytes.argtypes = [C.c_void_p, C.c_int]; lib.hipk_csr_product_bytes.restype = C.c_double
ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dt = F.HIPK_F64


def timeit(fn):
    for _ in range(5): fn()
    lib.hipk_sync(ctx)
    ms = C.c_float()
    lib.hipk_timer_start(ctx)
    for _ in range(reps): fn()
    lib.hipk_timer_stop(ctx, C.byref(ms))
    return 1e3 * ms.value / reps


for name, dims in (("lap2d_10m", (3162, 3163)), ("lap3d_2m", (125, 126, 127))):
    rp, ci, va, n = problems.laplacian_csr(dims)
    nnz = len(va)
    A = C.c_void_p()
    assert lib.hipk_csr_create(ctx, dt, n, n, 0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(A)) == 0
    # panels large enough that consecutive products do not find their vectors in the 256 MiB Infinity Cache: rotate over them
    ncol = max(2, int(3.0e9 // (8 * n)))
    X = torch.randn((ncol, n), dtype=torch.float64, device="cuda"); Y = torch.zeros((ncol, n), dtype=torch.float64, device="cuda")
    XO = torch.zeros((ncol, n), dtype=torch.float64, device="cuda")
    red = torch.zeros(64, dtype=torch.float64, device="cuda"); nn = torch.tensor([float(n)], dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    print(f"{name}: n = {n}, nnz = {nnz}, patterns = {lib.hipk_csr_npatterns(A)}, vectors rotate over {ncol} columns")
    ys = {}
    for fmt in (0, 1):
        lib.hipk_set_spmv_format(fmt)
        f = lib.hipk_csr_format(A)
        it = [0]
        def plain():
            c = it[0] % ncol; it[0] += 1
            lib.hipk_csr_matvec(A, None, X[c].data_ptr(), n, Y[c].data_ptr(), n, 1)
        def fused():
            c = it[0] % ncol; it[0] += 1
            lib.hipk_csr_matvec_scaled(A, ctx, X[c].data_ptr(), nn.data_ptr(), XO[c].data_ptr(), Y[c].data_ptr(), red.data_ptr())
        for label, fn, fz in (("plain", plain, 0), ("fused tail", fused, 1)):
            us = timeit(fn)
            alg = nnz * 12 + (n + 1) * 4 + (3 if fz else 2) * n * 8
            real = lib.hipk_csr_product_bytes(A, fz)
            print(f"  format {f} ({'row patterns' if f == 2 else 'CSR row tiles'}) {label:11s} {us:8.1f} us   {alg / us / 1e3:7.0f} GB/s of plain-CSR bytes   "
                  f"{real / us / 1e3:7.0f} GB/s of the {real / 1e6:.0f} MB it moves = {real / us / 1e3 / 8000:.3f} of 8 TB/s")
        lib.hipk_csr_matvec(A, None, X[0].data_ptr(), n, Y[0].data_ptr(), n, 1)
        lib.hipk_csr_matvec_scaled(A, ctx, X[1].data_ptr(), nn.data_ptr(), XO[1].data_ptr(), Y[1].data_ptr(), red.data_ptr())
        lib.hipk_sync(ctx); torch.cuda.synchronize()
        ys[fmt] = (Y[0].cpu().numpy().copy(), Y[1].cpu().numpy().copy(), XO[1].cpu().numpy().copy(), float(red[0].cpu()))
    same = all(np.array_equal(ys[0][i], ys[1][i]) for i in range(3))
    print(f"  y / fused y / normalised vector bit-identical between the forms: {same};  t'At tiles {ys[0][3]!r} patterns {ys[1][3]!r}")
    lib.hipk_set_spmv_format(1)
    lib.hipk_csr_destroy(A)
    del X, Y, XO
    torch.cuda.empty_cache()
g = C.c_double()
lib.hipk_read_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
lib.hipk_bandwidth_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
lib.hipk_read_probe(ctx, 2 << 30, 10, C.byref(g)); print("read probe", round(g.value, 1), "GB/s")
lib.hipk_bandwidth_probe(ctx, 1 << 30, 10, C.byref(g)); print("copy probe (read+write)", round(g.value, 1), "GB/s")
lib.hipk_ctx_destroy(ctx)
