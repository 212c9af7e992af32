"""BASELINE configs[3] on one GPU: complex Hermitian band matrix n = 4 M (half-bandwidth 3),
6 largest, block size 4, GD+k, basis 20 / restart 8, through hip_zprimme: natively on complex panels (default) or,
with FORM=real, on the real-equivalent form (csrc/eigs_complex.c).  Writes gpurun_out/config4_run_<form>.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems
from checkers import Operator, Session

n = int(os.environ.get("N", 4_000_000))
rp, ci, va = problems.hermitian_banded_csr(n)
form = os.environ.get("FORM", "native")          # native complex panels (default) | real: the real-equivalent form
s = Session(Operator(n, csr=(rp, ci, va)), dtype=np.complex128, backend="hip", complex_form=form)
out = []
for rep in range(2):
    t = time.time()
    r = s.solve(numEvals=6, target="largest", eps=1e-8, maxBlockSize=4, maxBasisSize=20, minRestartSize=8,
                method="GD_plusK", iseed=(2, 3, 5, 7), profile=(rep == 1))
    dt = time.time() - t
    AX = problems.csr_matvec_numpy(rp, ci, va, r.evecs) if rep == 0 else None
    out.append(dict(rep=rep, ret=r.ret, seconds=dt, evals=r.evals.tolist(), resNorms=r.resNorms.tolist(),
                    true_res=None if AX is None else np.linalg.norm(AX - r.evecs * r.evals, axis=0).tolist(),
                    matvecs=r.stats["numMatvecs"], outer=r.stats["numOuterIterations"], aNorm=r.params["aNorm"],
                    solver_seconds=r.stats["elapsedTime"], timeMatvec=r.stats["timeMatvec"], timeOrtho=r.stats["timeOrtho"]))
    print(out[-1], flush=True)
s.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(workload=f"configs[3] hermitian band n={n} hbw=3, 6 largest, b=4, GD+k, one GPU, {form} form", runs=out),
          open(f"gpurun_out/config4_run_{form}.json", "w"), indent=1)
