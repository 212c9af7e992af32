"""Generates tests/golden/reference_kernels.json (fixture F6 of SURVEY.md §8(c)): inputs and outputs of the
REFERENCE's own panel routines — update_projection_dprimme, Num_update_VWXR_dprimme, Bortho_gen_dprimme,
Bortho_block_dprimme — called by oracle/ref_kernel_harness.c, which is compiled against the reference's
headers where they lie and linked with oracle/_ref/libprimme_ref.so.  Run in the build container
(needs /root/reference: the harness is compiled against the reference's headers where they lie); the JSON is what the tests
read, here and on the GPU box.  (The reference's SOURCES never travel; the prebuilt oracle/_ref/libprimme_ref.so does — it is
git-ignored, not gpurun-ignored — and on the GPU box only bench.py's cpu_baseline leg loads it, in a process of its own.)

    python tests/golden/make_kernel_golden.py
"""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["make", "ref", "-j8"], cwd=os.path.join(ROOT, "oracle"))
subprocess.check_call(["make", "kernel-fixture"], cwd=os.path.join(ROOT, "oracle"))
out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_kernel_harness")], text=True)
data = json.loads(out)
# round 5: the same four routines at large shapes (inputs are closed forms the test regenerates; m-sized outputs as sums + samples)
WIDE = [(4099, 15, 8, 9), (4099, 41, 1, 0), (100003, 15, 1, 9), (100003, 41, 8, 9)]
data["wide"] = [json.loads(subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_kernel_harness"), "wide", *map(str, w)], text=True))
                for w in WIDE]
data["generated_by"] = "tests/golden/make_kernel_golden.py -> oracle/ref_kernel_harness.c (reference PRIMME, double, MKL BLAS)"
path = os.path.join(ROOT, "tests", "golden", "reference_kernels.json")
json.dump(data, open(path, "w"))
print("wrote", path, os.path.getsize(path), "bytes")
