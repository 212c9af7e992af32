"""Differential fuzz: random solver configurations on the HIP path against the same host code over
the plain-C kernel layer (tests' checker).  Catches device-layer limits the checker does not have
(jobs per launch, columns per launch, basis sizes).
   python scripts/fuzz_hip_vs_host.py N seed host FILE    # checker leg (CPU, slow): results -> FILE
   python scripts/fuzz_hip_vs_host.py N seed hip FILE     # HIP leg on the GPU box, compared with FILE
FUZZ_MASS=1 (both legs): every case is a generalised problem A x = lambda B x with B = problems.mass_matrix_csr, randomly scaled.
FUZZ_COMPLEX=1 (both legs): Hermitian problems on the native complex panels (problems.hermitian_banded_csr with a random diagonal
scaling, complex128 / complex64; with FUZZ_MASS the Hermitian mass matrix)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure
from primme_amd import problems
from checkers import Operator, eigsh

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
leg = sys.argv[3] if len(sys.argv) > 3 else "both"
ref_file = sys.argv[4] if len(sys.argv) > 4 else None
saved = json.load(open(ref_file)) if (leg == "hip" and ref_file) else {}
out_host = {}
rng = np.random.default_rng(seed)


class _Saved:
    def __init__(self, d):
        self.ret, self.initSize = d["ret"], d["initSize"]
        self.evals = np.array(d["evals"]); self.stats = dict(numOuterIterations=d["its"]); self.params = dict(aNorm=d["aNorm"])

METHODS = ["DYNAMIC", "DEFAULT_MIN_TIME", "DEFAULT_MIN_MATVECS", "Arnoldi", "GD", "GD_plusK", "GD_Olsen_plusK", "JD_Olsen_plusK",
           "RQI", "JDQR", "JDQMR", "JDQMR_ETol", "STEEPEST_DESCENT", "LOBPCG_OrthoBasis", "LOBPCG_OrthoBasis_Window"]
bad, ran, skipped = [], 0, 0
t0 = time.time()
for it in range(N):
    dims = tuple(int(x) for x in rng.integers(6, 15, size=int(rng.integers(2, 4))))
    rp, ci, va, n = problems.laplacian_csr(dims)
    va = va * (1.0 + 0.3 * np.sin(np.arange(len(va))) * (rng.random() < 0.5))   # sometimes not a Laplacian
    if rng.random() < 0.5:   # keep it symmetric: scale rows and columns alike
        d = 1.0 + 0.5 * rng.random(n)
        rows = np.repeat(np.arange(n), np.diff(rp))
        va = problems.laplacian_csr(dims)[2] * d[rows] * d[ci]
    else:
        va = problems.laplacian_csr(dims)[2]
    kw = dict(method=str(rng.choice(METHODS)), numEvals=int(min(rng.choice([1, 2, 3, 5, 8, 12, 20, 30]), max(1, n // 8))), eps=float(rng.choice([1e-6, 1e-9, 1e-11])),
              iseed=tuple(int(x) for x in rng.integers(0, 4000, 4)))
    dtype = np.float32 if rng.random() < 0.2 else np.float64
    if dtype == np.float32: kw["eps"] = 1e-4
    CPLX = bool(os.environ.get("FUZZ_COMPLEX"))
    if CPLX:
        n = int(rng.integers(60, 500))
        rp, ci, va = problems.hermitian_banded_csr(n, hbw=int(rng.integers(1, 5)))
        dsc = 1.0 + 0.5 * rng.random(n)
        rows_ = np.repeat(np.arange(n), np.diff(rp))
        va = va * dsc[rows_] * dsc[ci]
        kw["numEvals"] = int(min(kw["numEvals"], max(1, n // 8)))
        dtype = np.complex64 if dtype == np.float32 else np.complex128
        va = va.astype(dtype)
    target = str(rng.choice(["smallest", "largest", "closest_abs", "closest_geq", "closest_leq", "largest_abs"], p=[.35, .25, .15, .1, .1, .05]))
    kw["target"] = target
    if target.startswith("closest") or target == "largest_abs":
        kw["targetShifts"] = [float(rng.uniform(0.5, 6.0))]
    r = rng.random()
    # (blocks that are a sizeable fraction of the space, bases larger than the space: the reference itself
    #  does not return for some of these -- n = 88, block 20, 3 constraints, float restarts the same full
    #  basis forever; this solver returns -3 there, see the idle guards in eigs_main.c)
    if r < 0.5: kw["maxBlockSize"] = int(min(rng.choice([1, 2, 3, 4, 6, 8, 12, 20, 40]), max(1, n // 4)))
    if rng.random() < 0.4: kw["maxBasisSize"] = int(rng.choice([8, 12, 20, 40, 80, 150, 220]))
    if rng.random() < 0.3: kw["locking"] = int(rng.integers(0, 2))
    if rng.random() < 0.3: kw["precond"] = "jacobi" if rng.random() < 0.5 else ("jacobi", float(rng.uniform(-1, 1)))
    if rng.random() < 0.15 and dtype == np.float64 and kw.get("maxBlockSize", 1) == 1 and target.startswith("closest"):
        kw["projection"] = str(rng.choice(["harmonic", "refined"]))
    if rng.random() < 0.15:
        nc_ = int(rng.integers(1, 4))
        kw["constraints"] = np.linalg.qr(rng.standard_normal((n, nc_)) + (1j * rng.standard_normal((n, nc_)) if CPLX else 0.0))[0]
    kw["maxMatvecs"] = 15000
    B = None
    if os.environ.get("FUZZ_MASS"):
        brp, bci, bva = problems.hermitian_mass_matrix_csr(n) if CPLX else problems.mass_matrix_csr(n)
        bva = bva * float(rng.choice([1.0, 0.25, 7.0]))
        B = (brp, bci, bva.astype(dtype))
        kw.pop("projection", None)
        kw["mass"] = Operator(n, csr=B)
        kw["maxMatvecs"] = 40000
    try:
        if leg == "hip":
            if str(it) not in saved: continue
            h = _Saved(saved[str(it)])
        elif leg == "host":
            # each checker solve in a forked child with a wall-clock limit: some random configurations
            # make the reference algorithm itself spin (see the notes above), and the C code cannot be interrupted
            import multiprocessing as mp
            def child(conn):
                hh = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=dtype, **kw)
                conn.send(dict(ret=hh.ret, initSize=hh.initSize, evals=[float(x) for x in hh.evals], its=hh.stats["numOuterIterations"], aNorm=hh.params["aNorm"]))
            pc, cc = mp.Pipe()
            pr = mp.get_context("fork").Process(target=child, args=(cc,))
            pr.start()
            if pc.poll(25):
                out_host[str(it)] = pc.recv()
                pr.join()
            else:
                pr.kill(); pr.join()
                if os.environ.get("FUZZ_TRACE"): print(it, "checker did not return in 25 s: skipped", dims, {k: v for k, v in kw.items() if k != "constraints"}, flush=True)
                continue
            if os.environ.get("FUZZ_TRACE"): print(it, round(time.time() - t0, 1), dims, kw.get("method"), out_host[str(it)]["ret"], flush=True)
            continue
        else:
            h = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=dtype, **kw)
            out_host[str(it)] = dict(ret=h.ret, initSize=h.initSize, evals=[float(x) for x in h.evals], its=h.stats["numOuterIterations"], aNorm=h.params["aNorm"])
            if leg == "host":
                if os.environ.get("FUZZ_TRACE"): print(it, round(time.time() - t0, 1), dims, kw.get("method"), h.ret, flush=True)
                continue
        g = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", dtype=dtype, **kw)
    except Exception as e:
        bad.append(dict(it=it, err=repr(e)[:200], kw={k: (v if not isinstance(v, np.ndarray) else "array") for k, v in kw.items() if k != "mass"}))
        continue
    ran += 1
    if os.environ.get("FUZZ_TRACE"): print(it, round(time.time() - t0, 1), dims, kw.get("method"), h.ret, g.ret, flush=True)
    desc = dict(it=it, dims=dims, dtype=np.dtype(dtype).name, kw={k: (v if not isinstance(v, np.ndarray) else f"array{v.shape}") for k, v in kw.items() if k != "mass"},
                host=(h.ret, h.initSize, h.stats["numOuterIterations"]), hip=(g.ret, g.initSize, g.stats["numOuterIterations"]))
    if h.ret != g.ret and not (h.ret in (0, -3) and g.ret in (0, -3)):
        bad.append(dict(kind="ret", **desc)); continue
    if h.ret != 0 or g.ret != 0:
        skipped += 1; continue
    aN = max(h.params["aNorm"], 1e-300)
    single = dtype in (np.float32, np.complex64)
    tol = 2e-4 if single else 1e-8
    k = min(h.initSize, g.initSize)
    if h.initSize != g.initSize or (k and np.max(np.abs(np.sort(h.evals[:k]) - np.sort(g.evals[:k]))) > tol * aN):
        # interior targets may legitimately pick different members of a cluster: check residuals instead
        wide = np.complex128 if CPLX else np.float64
        X = g.evecs[:, :g.initSize].astype(wide)
        BX = X if B is None else problems.csr_matvec_numpy(B[0], B[1], B[2].astype(wide), X)
        R = problems.csr_matvec_numpy(rp, ci, va.astype(wide), X) - BX * g.evals[:g.initSize].astype(np.float64)
        rn = np.linalg.norm(R, axis=0)
        if g.initSize != kw["numEvals"] or np.any(rn > 10 * max(kw["eps"], np.finfo(np.float32 if single else np.float64).eps * 50) * aN):
            bad.append(dict(kind="values", maxres=float(rn.max()) if len(rn) else None, **desc))
if leg == "host":
    json.dump(out_host, open(ref_file, "w"))
    print(len(out_host), "checker results saved", round(time.time() - t0, 1), "s")
    sys.exit(0)
print(json.dumps(dict(ran=ran, not_converged_both=skipped, bad=len(bad), seconds=round(time.time() - t0, 1))))
for b in bad[:30]:
    print(json.dumps(b, default=str))
