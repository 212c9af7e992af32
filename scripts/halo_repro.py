"""Repeat the device leg of tests/test_kernels_gpu.py::test_csr_row_slabs_with_halo many times in ONE process and say,
for every mismatch, whether the kernel wrote a wrong value or the copy-back returned one (round-2 driver run: the
one-column halo product was off by 1e3 in interior rows once, on the same bytes that passed on another lease).

The output vector is pre-filled with NaN on the device (an unwritten row shows as NaN), the copy-back goes into a
pinned host buffer poisoned with 1e300 (a short copy shows as 1e300), a failing product is read back a second time,
recomputed into a second buffer on the same matrix handle, and recomputed on a freshly created handle.

  python scripts/halo_repro.py --iters 300            # uploads ordered on the context's stream (the fix)
  HIPK_LEGACY_UPLOAD=1 python scripts/halo_repro.py   # round-2 NULL-stream hipMemcpy / hipMemset uploads
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from primme_amd import _ffi as F, problems  # noqa: E402

CASES = [((60, 70), 2), ((23, 19, 17), 3), ((5000,), 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    import torch
    lib = F.load_product()
    dt = F.HIPK_F64 if args.dtype == "f64" else F.HIPK_F32
    npdt = np.float64 if dt == F.HIPK_F64 else np.float32
    tol = 1e-12 if dt == F.HIPK_F64 else 2e-5
    prepared = []
    for dims, nslabs in CASES:
        rng = np.random.default_rng(sum(dims) + nslabs)
        n = int(np.prod(dims))
        rp0, ci0, va0, _ = problems.laplacian_csr(dims)
        X = rng.standard_normal((3, n))
        Yref = problems.csr_matvec_numpy(rp0, ci0, va0, X.T).T
        base, rem = divmod(n, nslabs)
        for sidx in range(nslabs):
            nloc = base + (1 if sidx < rem else 0)
            row0 = sidx * base + min(sidx, rem)
            rp, ci, va, _ = problems.laplacian_csr(dims, row0=row0, nrows=nloc)
            prepared.append((dims, nslabs, sidx, n, nloc, row0, rp, ci, np.ascontiguousarray(va, dtype=npdt), X, Yref))

    def dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
        torch.cuda.synchronize()
        return t

    def ptr(t, off=0):
        return C.c_void_p(t.data_ptr() + off * t.element_size())

    pinned = torch.empty(3 * 6000, dtype=torch.float64 if dt == F.HIPK_F64 else torch.float32).pin_memory()

    def readback(ctx, t):
        lib.hipk_sync(ctx)
        torch.cuda.synchronize()
        flat = pinned[:t.numel()]
        flat.fill_(1e300 if dt == F.HIPK_F64 else 1e30)
        flat.copy_(t.reshape(-1))
        torch.cuda.synchronize()
        return flat.numpy().astype(np.float64).reshape(t.shape).copy()

    def create(ctx, nloc, n, row0, rp, ci, va):
        A = C.c_void_p()
        rc = lib.hipk_csr_create(ctx, dt, nloc, n, row0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                 va.ctypes.data_as(C.c_void_p), C.byref(A))
        assert rc == 0
        return A

    events = []
    nchecks = 0
    t0 = time.time()
    for it in range(args.iters):
        for (dims, nslabs, sidx, n, nloc, row0, rp, ci, va, X, Yref) in prepared:
            ctx = C.c_void_p()
            assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
            A = create(ctx, nloc, n, row0, rp, ci, va)
            lo, hi = int(lib.hipk_csr_halo_lo(A)), int(lib.hipk_csr_halo_hi(A))
            xl = dev(X[:, row0:row0 + nloc].astype(npdt))
            xlo = dev(np.ascontiguousarray(X[:, row0 - lo:row0]).astype(npdt) if lo else np.zeros((3, 1), npdt))
            xhi = dev(np.ascontiguousarray(X[:, row0 + nloc:row0 + nloc + hi]).astype(npdt) if hi else np.zeros((3, 1), npdt))
            assert lib.hipk_csr_set_halo_ld(A, ptr(xlo), max(lo, 1), ptr(xhi), max(hi, 1)) == 0
            y = dev(np.full((3, nloc), np.nan, npdt))
            assert lib.hipk_csr_matvec(A, None, ptr(xl), nloc, ptr(y), nloc, 3) == 0
            y3 = readback(ctx, y)
            y1 = dev(np.full((1, nloc), np.nan, npdt))
            assert lib.hipk_csr_matvec(A, None, ptr(xl), nloc, ptr(y1), nloc, 1) == 0
            y1v = readback(ctx, y1)
            red = dev(np.array([7.5, 0.0, 0.0]))
            xout = dev(np.full((1, nloc), np.nan, npdt)); yf = dev(np.full((1, nloc), np.nan, npdt))
            assert lib.hipk_csr_matvec_scaled(A, ctx, ptr(xl), ptr(red), ptr(xout), ptr(yf), ptr(red, 1)) == 0
            yfv = readback(ctx, yf)
            scale = max(1.0, np.abs(Yref).max())
            a = 1.0 / np.sqrt(7.5)
            want = Yref[:, row0:row0 + nloc]
            checks = (("spmm3", y3, want), ("spmv1", y1v, want[:1]), ("fused", yfv, a * want[:1]))
            for name, got, ref in checks:
                nchecks += 1
                err = np.abs(got - ref)
                bad = ~(err <= 50 * tol * scale)
                if not bad.any():
                    continue
                rows = np.nonzero(bad.any(axis=0))[0]
                ev = {"iter": it, "dims": dims, "slab": sidx, "kernel": name, "bad_rows": int(rows.size),
                      "first_bad": int(rows[0]), "last_bad": int(rows[-1]), "tiles_of_256": sorted({int(r) // 256 for r in rows})[:12],
                      "nan": int(np.isnan(got[bad]).sum()), "poison": int((np.abs(got[bad]) > 1e29).sum()),
                      "max_err": float(np.nanmax(err)), "sample": [float(v) for v in got[bad][:4]]}
                if name == "spmv1":
                    again = readback(ctx, y1)
                    ev["second_readback_same"] = bool(np.array_equal(again, y1v, equal_nan=True))
                    ev["second_readback_ok"] = bool((np.abs(again - ref) <= 50 * tol * scale).all())
                    y1b = dev(np.full((1, nloc), np.nan, npdt))
                    lib.hipk_csr_matvec(A, None, ptr(xl), nloc, ptr(y1b), nloc, 1)
                    ev["rerun_same_handle_ok"] = bool((np.abs(readback(ctx, y1b) - ref) <= 50 * tol * scale).all())
                    B = create(ctx, nloc, n, row0, rp, ci, va)
                    lib.hipk_csr_set_halo_ld(B, ptr(xlo), max(lo, 1), ptr(xhi), max(hi, 1))
                    y1c = dev(np.full((1, nloc), np.nan, npdt))
                    lib.hipk_csr_matvec(B, None, ptr(xl), nloc, ptr(y1c), nloc, 1)
                    ev["rerun_fresh_handle_ok"] = bool((np.abs(readback(ctx, y1c) - ref) <= 50 * tol * scale).all())
                    lib.hipk_csr_destroy(B)
                events.append(ev)
                print("MISMATCH", json.dumps(ev), flush=True)
            lib.hipk_csr_destroy(A)
            lib.hipk_ctx_destroy(ctx)
            del xl, xlo, xhi, y, y1, red, xout, yf
    print(json.dumps({"uploads": "legacy NULL stream" if os.environ.get("HIPK_LEGACY_UPLOAD") else "context stream",
                      "dtype": args.dtype, "iters": args.iters, "products_checked": nchecks, "mismatches": len(events),
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if events else 0


if __name__ == "__main__":
    sys.exit(main())
