"""Random GD+k / block-size-1 configurations (the ones the fused / speculative restart paths serve): iteration, matvec
and restart counts of the product host solver (hostcheck backend) against the live reference.  CPU only.
usage: python scripts/count_parity_sweep.py <seed> <cases>"""
import sys, os, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import checkers
from checkers import eigsh
from primme_amd import problems, Operator
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for t in range(N):
    dims = tuple(int(x) for x in rng.integers(6, 22, size=rng.integers(1, 4)))
    rp, ci, va, n = problems.laplacian_csr(dims)
    if n < 60: continue
    nev = int(rng.integers(1, 13))
    K = int(rng.integers(max(nev // 2 + 6, 8), 34))
    mr = int(rng.integers(2, max(3, K - 4)))
    kw = dict(numEvals=nev, eps=float(10.0 ** -rng.integers(6, 12)), aNorm=4.0 * len(dims), v0=problems.start_vector(n),
              method="GD_plusK", maxBasisSize=K, minRestartSize=mr, maxBlockSize=1, target=str(rng.choice(["smallest", "largest"])))
    if rng.random() < 0.5: kw["locking"] = int(rng.integers(0, 2))
    if rng.random() < 0.3: kw["maxPrevRetain"] = int(rng.integers(0, 4))
    try:
        h = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", **kw)
        r = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", **kw)
    except Exception as e:
        print("EXC", dims, kw, e); bad += 1; continue
    keys = ("numOuterIterations", "numMatvecs", "numRestarts")
    same = h.ret == r.ret and all(h.stats[k] == r.stats[k] for k in keys)
    ok_vals = h.ret != 0 or np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) <= 1e-9 * kw["aNorm"]
    if not (same and ok_vals):
        bad += 1
        print("DIFF", dims, {k: v for k, v in kw.items() if k != "v0"}, h.ret, r.ret, [h.stats[k] for k in keys], [r.stats[k] for k in keys])
print("cases", N, "differences", bad)
