"""Deterministic synthetic inputs of the benchmark configurations.

Everything is generated from integers and closed-form expressions, so the same
bytes are produced in this container and on the GPU box.
"""
import numpy as np


def laplacian_csr(dims, row0=0, nrows=None, dtype=np.float64):
    """CSR (int32 indices, global column numbers) of the 1/2/3-D Dirichlet Laplacian
    (diag 2*d, off-diagonals -1) on a grid dims = (nx,), (nx, ny) or (nx, ny, nz), x fastest.
    Returns rows [row0, row0+nrows)."""
    dims = tuple(int(d) for d in dims)
    nx = dims[0]
    ny = dims[1] if len(dims) > 1 else 1
    nz = dims[2] if len(dims) > 2 else 1
    n = nx * ny * nz
    if nrows is None:
        nrows = n - row0
    g = np.arange(row0, row0 + nrows, dtype=np.int64)
    ix = g % nx
    iy = (g // nx) % ny
    iz = g // (nx * ny)
    d = len([x for x in (nx, ny, nz) if x > 1]) if n > 1 else 1
    if len(dims) == 1:
        d = 1
    cols = []
    vals = []
    # ascending column order inside each row: -plane, -nx, -1, diag, +1, +nx, +plane
    cand = []
    if nz > 1:
        cand.append((g - nx * ny, iz > 0))
    if ny > 1:
        cand.append((g - nx, iy > 0))
    cand.append((g - 1, ix > 0))
    cand.append((g, np.ones_like(g, dtype=bool)))
    cand.append((g + 1, ix < nx - 1))
    if ny > 1:
        cand.append((g + nx, iy < ny - 1))
    if nz > 1:
        cand.append((g + nx * ny, iz < nz - 1))
    mask = np.stack([c[1] for c in cand], axis=1)
    col = np.stack([c[0] for c in cand], axis=1)
    diag_pos = [i for i, c in enumerate(cand) if c[0] is g][0]
    val = -np.ones(col.shape, dtype=dtype)
    val[:, diag_pos] = 2.0 * d
    counts = mask.sum(axis=1)
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    cols = col[mask].astype(np.int32)
    vals = val[mask].astype(dtype)
    assert rowptr[-1] < 2**31
    return rowptr.astype(np.int32), cols, vals, n


def laplacian_eigenvalues(dims, k):
    """k smallest analytic eigenvalues: sum_d 2 - 2cos(i_d pi/(n_d+1))."""
    axes = [2.0 - 2.0 * np.cos(np.arange(1, d + 1) * np.pi / (d + 1)) for d in dims if d >= 1]
    # only the low end of each axis can contribute to the k smallest sums
    axes = [a[: min(len(a), k + 2)] for a in axes]
    tot = axes[0]
    for a in axes[1:]:
        tot = np.add.outer(tot, a).ravel()
    return np.sort(tot)[:k]


def start_vector(n, row0=0, nrows=None, j=0, dtype=np.float64):
    """Deterministic initial guess v_i = sin(1 + i*0.6180339887 + j) (removes RNG dependence)."""
    if nrows is None:
        nrows = n - row0
    i = np.arange(row0, row0 + nrows, dtype=np.float64)
    return np.sin(1.0 + i * 0.6180339887498949 + 0.37 * j).astype(dtype)


def read_matrix_market(path):
    """Minimal Matrix-Market coordinate reader (real or complex; general / symmetric / Hermitian), returns scipy-free
    CSR (rowptr int32, colind int32, values float64, nrows, ncols).  Restates what the
    reference's test driver does with tests/COMMON/mmio.c + csr.c:46-265 (COO -> CSR,
    symmetric expansion)."""
    with open(path) as f:
        header = f.readline().lower().split()
        symmetric = "symmetric" in header
        hermitian = "hermitian" in header
        cplx = "complex" in header
        pattern = "pattern" in header
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nr, nc, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, ndmin=2)
    r = data[:, 0].astype(np.int64) - 1
    c = data[:, 1].astype(np.int64) - 1
    v = np.ones(len(r)) if pattern else data[:, 2].astype(np.float64)
    if cplx:
        v = data[:, 2] + 1j * data[:, 3]
    if symmetric or hermitian:
        off = r != c
        vm = np.conj(v[off]) if hermitian else v[off]
        r, c, v = np.concatenate([r, c[off]]), np.concatenate([c, r[off]]), np.concatenate([v, vm])
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    rowptr = np.zeros(nr + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), c.astype(np.int32), v, nr, nc


def tile_block_diagonal(rowptr, colind, values, ntiles, scale_fn=None, row0_tile=0):
    """Block-diagonal tiling of a square CSR matrix: tile t (global index row0_tile + t) is
    the matrix scaled by scale_fn(t) so that the spectrum stays simple."""
    n0 = len(rowptr) - 1
    nnz0 = len(values)
    rp = np.zeros(n0 * ntiles + 1, dtype=np.int64)
    ci = np.empty(nnz0 * ntiles, dtype=np.int64)
    va = np.empty(nnz0 * ntiles, dtype=values.dtype)
    base = rowptr.astype(np.int64)
    for t in range(ntiles):
        gt = row0_tile + t
        s = 1.0 if scale_fn is None else scale_fn(gt)
        rp[t * n0 + 1:(t + 1) * n0 + 1] = base[1:] + t * nnz0
        ci[t * nnz0:(t + 1) * nnz0] = colind.astype(np.int64) + gt * n0
        va[t * nnz0:(t + 1) * nnz0] = values * s
    assert ci.max() < 2**31 and rp[-1] < 2**31
    return rp.astype(np.int32), ci.astype(np.int32), va


def mass_matrix_csr(n, row0=0, nrows=None):
    """A closed-form SPD mass matrix B for generalised problems A x = lambda B x (tests): tridiagonal,
    diagonal 1 + 0.5 sin^2(0.1 i), off-diagonals 0.15 (strictly diagonally dominant)."""
    nrows = n if nrows is None else nrows
    rows, cols, vals = [], [], []
    for i in range(row0, row0 + nrows):
        for d in (-1, 0, 1):
            j = i + d
            if 0 <= j < n:
                rows.append(i - row0); cols.append(j)
                vals.append(1.0 + 0.5 * np.sin(0.1 * i) ** 2 if d == 0 else 0.15)
    rp = np.zeros(nrows + 1, dtype=np.int64)
    np.add.at(rp, np.array(rows) + 1, 1)
    return np.cumsum(rp).astype(np.int32), np.array(cols, dtype=np.int32), np.array(vals)


def hermitian_mass_matrix_csr(n):
    """Hermitian positive definite mass matrix for complex generalised problems (tests): tridiagonal, real diagonal
    1 + 0.5 sin^2(0.1 i), off-diagonals 0.15 exp(+-0.3 i)."""
    rp, ci, va = mass_matrix_csr(n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = va.astype(np.complex128)
    v[ci > rows] = 0.15 * np.exp(0.3j)
    v[ci < rows] = 0.15 * np.exp(-0.3j)
    return rp, ci, v


def csr_matvec_numpy(rowptr, colind, values, x):
    """y = A x for a CSR matrix with numpy (reference tests/COMMON/mat.c:64-90 amux)."""
    x = np.asarray(x)
    nrows = len(rowptr) - 1
    xr = x.reshape(x.shape[0], -1)
    rows = np.repeat(np.arange(nrows), np.diff(rowptr.astype(np.int64)))
    y = np.zeros((nrows, xr.shape[1]), dtype=np.result_type(values, xr))
    for c in range(xr.shape[1]):
        w = values * xr[colind, c]
        if np.iscomplexobj(w):
            y[:, c] = np.bincount(rows, weights=w.real, minlength=nrows) + 1j * np.bincount(rows, weights=w.imag, minlength=nrows)
        else:
            y[:, c] = np.bincount(rows, weights=w, minlength=nrows)
    return y.reshape((len(rowptr) - 1,) + x.shape[1:])


SVDS_PRIMES = (1, 7919, 104729, 1299709, 15485863)


def svds_synthetic_csr(m, n, row0=0, nrows=None):
    """BASELINE configs[4] matrix: row i has 5 nonzeros at columns
    (i*p_q + q) mod n, p = SVDS_PRIMES, values 1 + ((i+q) mod 13)/13; duplicates in a row are summed.
    Returns CSR (rowptr int32, colind int32, values) for rows [row0, row0+nrows)."""
    nrows = m - row0 if nrows is None else nrows
    i = np.arange(row0, row0 + nrows, dtype=np.int64)
    cols = np.stack([(i * p + q) % n for q, p in enumerate(SVDS_PRIMES)], axis=1)
    vals = np.stack([1.0 + ((i + q) % 13) / 13.0 for q in range(5)], axis=1)
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    vals = np.take_along_axis(vals, order, axis=1)
    # merge duplicate columns inside a row
    dup = np.zeros_like(cols, dtype=bool)
    dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
    if dup.any():
        for q in range(4, 0, -1):
            d = dup[:, q]
            vals[d, q - 1] += vals[d, q]
    keep = ~dup
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(keep.sum(axis=1))
    return rowptr.astype(np.int32), cols[keep].astype(np.int32), vals[keep]


def hermitian_graded_csr(n, hbw=3):
    """Closed-form complex Hermitian band matrix with a graded diagonal (well separated extremal eigenvalues: the
    parity fixtures of the native complex path): d_j = 0.05 (j + 1) + sin(j), a_{j,j+q} = 0.3 exp(0.37 i q (1 + j mod 3)) / (q + 1)."""
    j = np.arange(n, dtype=np.int64)
    offs = np.arange(-hbw, hbw + 1)
    cols = j[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < n)
    vals = np.zeros(cols.shape, dtype=np.complex128)
    for t, o in enumerate(offs):
        q = abs(int(o))
        lo = np.minimum(j, j + o)                      # the entry belongs to the pair (lo, lo + q)
        ph = np.exp(0.37j * q * (1 + lo % 3)) * 0.3 / (q + 1.0)
        vals[:, t] = ph if o > 0 else (np.conj(ph) if o < 0 else 0.05 * (j + 1) + np.sin(j))
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(ok.sum(axis=1))
    return rp.astype(np.int32), cols[ok].astype(np.int32), vals[ok]


def complex_start_vector(n):
    """start_vector with a closed-form phase per entry (complex parity runs)."""
    return start_vector(n) * np.exp(0.1j * np.arange(n))


def hermitian_banded_csr(n, row0=0, nrows=None, hbw=3):
    """BASELINE configs[3]: complex Hermitian band matrix, diagonal
    d_j = 2 + (j mod 97)/97, off-diagonals a_{j,j+q} = exp(0.37 i q)/(q+1), q = 1..hbw.
    Rows [row0, row0+nrows) with global column numbers; returns (rowptr, colind, values complex128)."""
    nrows = n - row0 if nrows is None else nrows
    j = np.arange(row0, row0 + nrows, dtype=np.int64)
    offs = np.arange(-hbw, hbw + 1)
    cols = j[:, None] + offs[None, :]
    ok = (cols >= 0) & (cols < n)
    q = np.abs(offs)
    band = np.where(offs > 0, np.exp(0.37j * q) / (q + 1.0), np.exp(-0.37j * q) / (q + 1.0))
    vals = np.broadcast_to(band[None, :], cols.shape).copy()
    vals[:, hbw] = 2.0 + (j % 97) / 97.0
    rp = np.zeros(nrows + 1, dtype=np.int64)
    rp[1:] = np.cumsum(ok.sum(axis=1))
    return rp.astype(np.int32), cols[ok].astype(np.int32), vals[ok]


def hermitian_tridiag_graded(n):
    """examples/ex_eigs_zhip_precond.hip: A = tridiag(conj(a), d_j, a), d_j = 1 + j, a = -0.5 exp(0.7 i)
    (unitarily similar to the real tridiag(|a|, d_j, |a|)).  Returns (rowptr, colind, values complex128, d, a)."""
    j = np.arange(n, dtype=np.int64)
    d = 1.0 + j
    a = -0.5 * np.cos(0.7) - 0.5j * np.sin(0.7)
    rows, cols, vals = [], [], []
    cnt = np.full(n, 3, dtype=np.int64); cnt[0] -= 1; cnt[-1] -= 1
    rp = np.zeros(n + 1, dtype=np.int64); rp[1:] = np.cumsum(cnt)
    ci = np.empty(rp[-1], dtype=np.int64); va = np.empty(rp[-1], dtype=np.complex128)
    pos = rp[:-1].copy()
    lo = j > 0
    ci[pos[lo]] = j[lo] - 1; va[pos[lo]] = np.conj(a); pos[lo] += 1
    ci[pos] = j; va[pos] = d; pos += 1
    hi = j < n - 1
    ci[pos[hi]] = j[hi] + 1; va[pos[hi]] = a
    return rp.astype(np.int32), ci.astype(np.int32), va, d, a


def rational_complex_start_vector(n):
    """start vector made of exactly representable quotients (the C example forms the same bits)"""
    j = np.arange(n, dtype=np.int64)
    return (((j * 7 + 3) % 11 - 5) / 5.0 + 1j * (((j * 5 + 1) % 13 - 6) / 6.0)).reshape(n, 1)


def zjacobi_rotation(n, gamma):
    """1 + i gamma w_j, w_j = ((j mod 7) - 3)/3: the non-Hermitian factor of the example's diagonal preconditioner"""
    j = np.arange(n, dtype=np.int64)
    return 1.0 + 1j * gamma * ((j % 7 - 3) / 3.0)
