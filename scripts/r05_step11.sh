#!/bin/bash
# round 5, step 11: native complex singular values on the device + timing of both forms
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step11; mkdir -p $O
timeout 900 python -m pytest tests/test_svds_gpu.py -m gpu -q -x -p no:cacheprovider > $O/svds_tests.txt 2>&1; echo "svds tests exit $?"; tail -3 $O/svds_tests.txt
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/complex_svds_forms.txt
import os, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from primme_amd import problems
from checkers import svds
m, n, k = 400000, 100000, 5
rp, ci, va = problems.svds_synthetic_csr(m, n)
rng = np.random.default_rng(5)
vz = va.astype(np.complex128) * np.exp(1j * rng.uniform(0, 2 * np.pi, size=len(va)))
for form in ("native", "real_equivalent"):
    if form == "real_equivalent": os.environ["PRIMME_AMD_COMPLEX_REAL_FORM"] = "1"
    else: os.environ.pop("PRIMME_AMD_COMPLEX_REAL_FORM", None)
    for rep in range(2):
        t0 = time.perf_counter()
        r = svds(m, n, (rp, ci, vz), numSvals=k, eps=1e-8, method="normalequations", backend="hip", dtype=np.complex128)
        dt = time.perf_counter() - t0
    print(f"{form:16s} ret {r.ret}  {dt:7.3f} s incl. set-up  matvecs {r.stats['numMatvecs']}  outer {r.stats['numOuterIterations']}  svals {np.round(r.svals[:3], 6)}  max resNorm/|A| {np.max(r.resNorms) / r.params['aNorm']:.2e}")
PY
