"""Random configurations of the NATIVE complex path (hip_zprimme's host solver over the CPU kernels, hostcheck backend)
against the live zprimme of the reference: iteration, matvec and restart counts.  Extremal targets, no preconditioner,
random Hermitian band matrices, methods of the Generalized-Davidson family and JDQMR, block sizes 1-4, locking on/off.
CPU only.   usage: python scripts/count_parity_sweep_complex.py <seed> <cases> [out.txt]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from checkers import eigsh, Operator
from test_complex_host import hermitian_band
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = []
bad = exact = 0
for t in range(N):
    n = int(rng.integers(200, 900))
    A, csr = hermitian_band(n, seed=int(rng.integers(1, 10**6)))
    nev = int(rng.integers(1, 9))
    b = int(rng.choice([1, 1, 2, 4]))
    method = str(rng.choice(["GD_plusK", "GD_plusK", "GD", "GD_Olsen_plusK", "JDQMR", "JDQMR_ETol", "LOBPCG_OrthoBasis"]))
    if method.startswith("JDQMR"): b = 1                       # block JDQMR is a different block iteration
    kw = dict(numEvals=nev, eps=float(10.0 ** -rng.integers(6, 11)), method=method, maxBlockSize=b,
              target=str(rng.choice(["smallest", "largest"])), iseed=tuple(int(x) for x in rng.integers(1, 4000, 4) | 1))
    if method == "LOBPCG_OrthoBasis": kw["maxBlockSize"] = nev
    if rng.random() < 0.5: kw["locking"] = int(rng.integers(0, 2))
    if rng.random() < 0.4 and method != "LOBPCG_OrthoBasis":
        K = int(rng.integers(max(nev + 8, 3 * b + 6), 40)); kw["maxBasisSize"] = K; kw["minRestartSize"] = int(rng.integers(max(2, b), max(3, K // 2)))
    try:
        h = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, **kw)
        r = eigsh(Operator(n, csr=csr), backend="reference", dtype=np.complex128, **kw)
    except Exception as e:
        out.append(f"EXC n={n} {kw} {e}"); bad += 1; continue
    keys = ("numOuterIterations", "numMatvecs", "numRestarts")
    same = h.ret == r.ret and all(h.stats[k] == r.stats[k] for k in keys)
    ok_vals = h.ret != 0 or np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) <= 1e-9 * h.params["aNorm"]
    exact += bool(same and ok_vals)
    if not (same and ok_vals):
        bad += 1
        out.append(f"DIFF n={n} {kw} ret {h.ret}/{r.ret} ours {[h.stats[k] for k in keys]} zprimme {[r.stats[k] for k in keys]} same eigenvalues {bool(ok_vals)}")
out.append(f"cases {N}: count-exact with the reference's eigenvalues {exact}, differences {bad}")
print("\n".join(out))
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("# native complex host solver (CPU kernels) vs live zprimme, scripts/count_parity_sweep_complex.py seed %s\n" % sys.argv[1] + "\n".join(out) + "\n")
