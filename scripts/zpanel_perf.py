"""Isolated launches of the complex TN panel product [Q V]^H X (hipk_panel_dots, HIPK_C64) at the shapes of BASELINE configs[3]
(m = 4 M complex rows, 4 right-hand columns): the vector-unit kernel (zdots_kernel) against the matrix-core one
(zdots_mfma_kernel, HIPK_ZMFMA=1), and the real kernels at the same byte counts for comparison.
usage: [HIPK_ZMFMA=1] [HIPK_ZDOTS_BPC=n] [HIPK_ZMFMA_BPC=n] python scripts/zpanel_perf.py [reps]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from primme_amd import _ffi as F
lib = F.load_product()
ctx = C.c_void_p(); assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = int(os.environ.get("KP_M", "4000000"))
V = torch.randn((28, m), dtype=torch.complex128, device="cuda")
red = torch.zeros(8192, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
tag = " ".join(f"{k}={os.environ[k]}" for k in ("HIPK_ZMFMA", "HIPK_ZDOTS_BPC", "HIPK_ZMFMA_BPC", "HIPK_MFMA_BPC", "HIPK_NO_MFMA") if k in os.environ) or "default"
def timeit(fn, nbytes, label):
    for _ in range(3): fn()
    lib.hipk_sync(ctx)
    ms = C.c_float()
    lib.hipk_timer_start(ctx)
    for _ in range(reps): fn()
    lib.hipk_timer_stop(ctx, C.byref(ms))
    us = 1e3 * ms.value / reps
    print(f"[{tag}] {label:44s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
for tot in (8, 12, 16, 20, 24):
    for nx in (4,):
        s = (F.HipkSeg * 3)()
        s[0].base, s[0].ld, s[0].ncols = V.data_ptr(), m, tot
        x = V[tot:tot + nx]
        timeit(lambda: lib.hipk_panel_dots(ctx, F.HIPK_C64, m, s, 1, x.data_ptr(), m, nx, red.data_ptr(), tot), (tot + nx) * m * 16, f"complex [V]^H X  tot={tot} nx={nx}")
# the real kernels on the same bytes (2m real rows)
Vr = torch.view_as_real(V).reshape(28, 2 * m)
for tot in (8, 16, 24):
    s = (F.HipkSeg * 3)()
    s[0].base, s[0].ld, s[0].ncols = Vr.data_ptr(), 2 * m, tot
    x = Vr[tot:tot + 4]
    timeit(lambda: lib.hipk_panel_dots(ctx, F.HIPK_F64, 2 * m, s, 1, x.data_ptr(), 2 * m, 4, red.data_ptr(), tot), (tot + 4) * m * 16, f"real    [V]' X   tot={tot} nx=4 (2m rows)")
lib.hipk_ctx_destroy(ctx)
