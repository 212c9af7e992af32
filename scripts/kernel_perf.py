"""Per-kernel achieved bandwidth at the config-2 shape (m = 2 000 250, f64), isolated launches.
usage: python scripts/kernel_perf.py [reps]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from primme_amd import _ffi as F, problems
lib = F.load_product()
ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = int(os.environ.get("KP_M", "2000250")); K = 16; L = 10
ld = (m + 15) // 16 * 16 if os.environ.get("KP_LDPAD") else m          # columns on 128-byte boundaries
dt = F.HIPK_F64
V = torch.randn((26, ld), dtype=torch.float64, device="cuda")
W = torch.randn((26, ld), dtype=torch.float64, device="cuda")
Q = torch.randn((L, ld), dtype=torch.float64, device="cuda")
red = torch.zeros(4096, dtype=torch.float64, device="cuda")
h = torch.randn((K, K), dtype=torch.float64, device="cuda")
th = torch.randn(K, dtype=torch.float64, device="cuda")
hhost = np.random.randn(32)
torch.cuda.synchronize()
def timeit(fn, nbytes, label):
    for _ in range(3): fn()
    lib.hipk_sync(ctx)
    ms = C.c_float()
    lib.hipk_timer_start(ctx)
    for _ in range(reps): fn()
    lib.hipk_timer_stop(ctx, C.byref(ms))
    us = 1e3 * ms.value / reps
    print(f"{label:50s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s")
def seg(t, n): 
    a = (F.HipkSeg * 3)(); a[0].base = t.data_ptr(); a[0].ld = ld; a[0].ncols = n; return a
for k in (6, 10, 15):
    s = (F.HipkSeg * 3)()
    s[0].base, s[0].ld, s[0].ncols = V.data_ptr(), ld, k
    s[1].base, s[1].ld, s[1].ncols = Q.data_ptr(), ld, L
    x = V[k]
    s[2].base, s[2].ld, s[2].ncols = x.data_ptr(), ld, 1
    timeit(lambda: lib.hipk_panel_dots(ctx, dt, m, s, 3, x.data_ptr(), ld, 1, red.data_ptr(), k + L + 1), (k + L + 2) * m * 8, f"dots CGS k={k} L={L}")
    s2 = (F.HipkSeg * 3)(); s2[0].base, s2[0].ld, s2[0].ncols = V.data_ptr(), ld, k + 1
    timeit(lambda: lib.hipk_panel_dots(ctx, dt, m, s2, 1, W[k].data_ptr(), ld, 1, red.data_ptr(), k + 1), (k + 2) * m * 8, f"dots proj k={k}")
    timeit(lambda: lib.hipk_panel_project(ctx, dt, m, s, 2, red.data_ptr(), k + L, x.data_ptr(), ld, 1, red.data_ptr() + 8 * 100), (k + L + 2) * m * 8, f"project k={k} L={L}")
    jobs = (F.HipkJob * 1)(); jobs[0].kind = F.HIPK_JOB_RES; jobs[0].col = 0; jobs[0].dst = V[k].data_ptr(); jobs[0].slot = 0
    timeit(lambda: lib.hipk_ritz_update(ctx, dt, m, V.data_ptr(), W.data_ptr(), ld, k, h.data_ptr(), K, th.data_ptr(), jobs, 1, red.data_ptr()), (2 * k + 1) * m * 8, f"ritz RES k={k}")
    timeit(lambda: lib.hipk_ritz_residual_overlaps(ctx, dt, m, V.data_ptr(), W.data_ptr(), ld, k, hhost.ctypes.data_as(C.c_void_p), C.c_double(0.3), V[k].data_ptr(), Q.data_ptr(), ld, L, 0, red.data_ptr()), (2 * k + L + 1) * m * 8, f"ritz+overlaps k={k} L={L}")
    timeit(lambda: lib.hipk_ritz_residual_overlaps(ctx, dt, m, V.data_ptr(), W.data_ptr(), ld, k, hhost.ctypes.data_as(C.c_void_p), C.c_double(0.3), V[k].data_ptr(), Q.data_ptr(), ld, L, 1, red.data_ptr()), (2 * k + L + 1) * m * 8, f"ritz+overlaps+W'r k={k} L={L}")
    timeit(lambda: lib.hipk_pair_dots(ctx, dt, m, V[k].data_ptr(), ld, W[k].data_ptr(), ld, 1, red.data_ptr()), 2 * m * 8, "pair dot t'w")
# the restart of configs[1]: basis 15 -> 8 columns of V and W, residual of the next candidate
def restart_jobs(rs, k, out_of_place):
    off = 16 if out_of_place else 0
    jb = (F.HipkJob * (2 * rs + 1))()
    for c in range(rs):
        jb[c].kind, jb[c].col, jb[c].dst, jb[c].slot = F.HIPK_JOB_XV, c, V[off + c].data_ptr(), -1
        jb[rs + c].kind, jb[rs + c].col, jb[rs + c].dst, jb[rs + c].slot = F.HIPK_JOB_XW, c, W[off + c].data_ptr(), -1
    jb[2 * rs].kind, jb[2 * rs].col, jb[2 * rs].dst, jb[2 * rs].slot = F.HIPK_JOB_RES, rs, V[off + rs].data_ptr(), 0
    return jb
if V.shape[0] >= 26:
    for (k, rs) in ((15, 8),):
        jb = restart_jobs(rs, k, False)
        timeit(lambda: lib.hipk_ritz_update(ctx, dt, m, V.data_ptr(), W.data_ptr(), ld, k, h.data_ptr(), K, th.data_ptr(), jb, 2 * rs + 1, red.data_ptr()), (2 * k + 2 * rs + 1) * m * 8, f"restart pass k={k} -> {rs} (+ residual)")
        for LL in (0, 5, 10):
            for oop in (False, True):
                jb2 = restart_jobs(rs, k, oop)
                timeit(lambda: lib.hipk_ritz_update_overlaps(ctx, dt, m, V.data_ptr(), W.data_ptr(), ld, k, h.data_ptr(), K, th.data_ptr(), jb2, 2 * rs + 1, red.data_ptr(), rs, Q.data_ptr(), ld, LL, red.data_ptr() + 8 * 64), (2 * k + LL + 2 * rs + 1) * m * 8, f"restart pass + overlaps k={k} -> {rs} L={LL} {'out of place' if oop else 'in place'}")
rp, ci, va, n = problems.laplacian_csr((125, 126, 127))
A = C.c_void_p()
assert lib.hipk_csr_create(ctx, dt, n, n, 0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(A)) == 0
timeit(lambda: lib.hipk_csr_matvec(A, None, V[0].data_ptr(), ld, W[0].data_ptr(), ld, 1), len(va) * 12 + (n + 1) * 4 + 2 * n * 8, "csr spmv 7pt")
nn = torch.tensor([float(m)], dtype=torch.float64, device="cuda")
timeit(lambda: lib.hipk_csr_matvec_scaled(A, ctx, V[0].data_ptr(), nn.data_ptr(), V[1].data_ptr(), W[0].data_ptr(), red.data_ptr()), len(va) * 12 + (n + 1) * 4 + 3 * n * 8, "csr spmv 7pt fused tail (scale + A t + t'At)")
S = C.c_void_p(); lib.hipk_stencil_create(ctx, dt, 125, 126, 127, 0, n, C.byref(S))
timeit(lambda: lib.hipk_csr_matvec(S, None, V[0].data_ptr(), ld, W[0].data_ptr(), ld, 1), 2 * n * 8, "stencil 7pt")
a1 = (C.c_double * 1)(0.5)
timeit(lambda: lib.hipk_scale_cols(ctx, dt, m, V[3].data_ptr(), ld, 1, a1), 2 * m * 8, "scale")
g = C.c_double(); lib.hipk_bandwidth_probe(ctx, 1 << 30, 10, C.byref(g)); print("copy probe (1 GiB, read+write)", g.value, "GB/s")
