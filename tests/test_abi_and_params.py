"""ABI and parameter-default parity with the reference (fixtures: tests/golden/reference_abi.json,
captured from the reference build by tests/golden/make_golden.py)."""
import ctypes as C
import json
import os

import pytest

from primme_amd import _ffi as F

import checkers

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_abi.json")))

# offsets probed from the reference headers with gcc
SURVEY_OFFSETS = {"n": 0, "matrixMatvec": 8, "numProcs": 52, "procID": 56, "nLocal": 64, "commInfo": 72,
                  "globalSumReal": 80, "broadcastReal": 96, "numEvals": 108, "target": 112, "targetShifts": 120,
                  "locking": 132, "initSize": 136, "maxBasisSize": 144, "minRestartSize": 148, "maxBlockSize": 152,
                  "maxMatvecs": 160, "iseed": 176, "aNorm": 208, "eps": 232, "orth": 240, "printLevel": 248,
                  "outputFile": 256, "matrix": 264, "ldevecs": 304, "ldOPs": 312, "correctionParams": 328,
                  "stats": 376, "convTestFun": 576, "monitorFun": 600, "queue": 624, "profile": 632}


def test_struct_layout_matches_reference():
    assert C.sizeof(F.PrimmeParams) == 640
    assert C.sizeof(F.PrimmeStats) == 200
    for name, off in SURVEY_OFFSETS.items():
        assert getattr(F.PrimmeParams, name).offset == off, name


def test_c_header_layout_matches_ctypes(built, tmp_path):
    """include/primme_amd.h compiled by gcc gives the same sizeof/offsetof table."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "probe.c"
    fields = list(SURVEY_OFFSETS)
    body = "\n".join(f'printf("{f} %zu\\n", offsetof(primme_params, {f}));' for f in fields)
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "primme_amd.h"\nint main(){printf("size %zu %zu\\n", sizeof(primme_params), sizeof(primme_stats));\n' + body + "\nreturn 0;}\n")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    assert out[0] == "size 640 200"
    got = dict(line.split() for line in out[1:] if line)
    for f in fields:
        assert int(got[f]) == SURVEY_OFFSETS[f], f


def _lib():
    return checkers.load_hostcheck()   # same eigs_params.c as the product library, loadable without a GPU


def test_initialize_bytes_identical_to_reference(built):
    p = F.PrimmeParams()
    C.memset(C.byref(p), 0xAB, C.sizeof(p))
    _lib().primme_initialize(C.byref(p))
    q = F.PrimmeParams.from_buffer_copy(bytes.fromhex(GOLD["initialize_bytes_hex"]))

    def same(x, y, path):
        if isinstance(x, C.Structure):
            for name, _ in x._fields_:
                if name == "outputFile":      # the process's own `stdout` pointer
                    continue
                same(getattr(x, name), getattr(y, name), path + "." + name)
        elif isinstance(x, C.Array):
            assert list(x) == list(y), path
        elif isinstance(x, (int, float, bytes)) or x is None:
            assert x == y or (not x and not y), path
        else:                                   # typed pointers: both NULL
            assert not bool(x) and not bool(y), path

    same(p, q, "primme")


@pytest.mark.parametrize("key", sorted(GOLD["presets"]))
def test_set_method_matches_reference(built, key):
    mname, nev, bs, tgt, prec = key.split("|")
    lib = _lib()
    p = F.PrimmeParams()
    lib.primme_initialize(C.byref(p))
    p.n = 10000
    p.numEvals = int(nev)
    p.maxBlockSize = int(bs)
    p.target = int(tgt)
    if int(prec):
        p.applyPreconditioner = 1
    rc = lib.primme_set_method(F.METHODS[mname], C.byref(p))
    g = GOLD["presets"][key]
    got = dict(rc=rc, maxBasisSize=p.maxBasisSize, minRestartSize=p.minRestartSize, maxBlockSize=p.maxBlockSize,
               locking=p.locking, dynamicMethodSwitch=p.dynamicMethodSwitch, maxPrevRetain=p.restartingParams.maxPrevRetain,
               precondition=p.correctionParams.precondition, robustShifts=p.correctionParams.robustShifts,
               maxInnerIterations=p.correctionParams.maxInnerIterations,
               projectors=[getattr(p.correctionParams.projectors, k) for k in ("LeftQ", "LeftX", "RightQ", "RightX", "SkewQ", "SkewX")],
               convTest=p.correctionParams.convTest, relTolBase=p.correctionParams.relTolBase,
               projection=p.projectionParams.projection, initBasisMode=p.initBasisMode)
    assert got == g


def test_library_exports_every_declared_symbol(built):
    """The C-ABI shared library loads and exports what include/*.h declare (no compute calls)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in sorted(os.listdir(os.path.join(root, "include"))):
        txt = open(os.path.join(root, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b((?:hipk|primme|hip|Num)_\w+|[dszc]primme)\s*\(", txt):
            declared.add(m.group(1))
    declared -= {"primme_block_op", "primme_svds_block_op"}
    syms = subprocess.check_output(["nm", "-D", "--defined-only", F.PRODUCT_LIB], text=True)
    exported = {line.split()[-1] for line in syms.splitlines() if line.strip()}
    missing = sorted(d for d in declared if d not in exported)
    assert not missing, missing
    # ... and nothing but the C ABI: the host solver's internals (pa_*) and the C++ helpers stay local (csrc/exports.map)
    leaked = sorted(e for e in exported if e.startswith("pa_") or e.startswith("_Z"))
    assert not leaked, leaked[:10]
    C.CDLL(F.PRODUCT_LIB)   # loads (HIP runtime present in the image even without a GPU)


def test_check_input_codes(built):
    lib = _lib()
    import numpy as np
    p = F.PrimmeParams()
    lib.primme_initialize(C.byref(p))
    ev = np.zeros(4); rn = np.zeros(4); vec = np.zeros((4, 10))
    p.n = 10
    p.numEvals = 4
    # no matvec -> -7 (reference primme_c.c:449)
    assert lib.hip_dprimme(ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -7
    p.matrixMatvec = 1
    p.numEvals = 11
    assert lib.hip_dprimme(ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -10
    p.numEvals = 4
    assert lib.hip_dprimme(None, vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -30
    p.target = F.primme_closest_abs
    assert lib.hip_dprimme(ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -14


def test_reference_named_backend_shim_on_the_cpu_checker(built, tmp_path):
    """include/primme_amd_wrapper.h (Num_gemm_ddh / Num_gemm_dhd / ... for the double GPU instantiation):
    the C test program of examples/ compiled against the CPU checker build runs every routine against
    plain loops (the same program runs against libprimme_amd.so in tests/test_c_examples_gpu.py)."""
    import subprocess
    exe = str(tmp_path / "test_hip_wrapper_cpu")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(root, "oracle", "_build")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "test_hip_wrapper.c"), "-o", exe,
                           "-L" + build, "-lprimme_hostcheck", "-Wl,-rpath," + build, "-lm"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0 and "returned 0" in out.stdout, out.stdout
