#!/bin/bash
# The whole `-m gpu` suite on the real device with the HOST code of the product library (the host side of every .hip file
# and every .c file) and of the checker under AddressSanitizer (scripts/build_hostasan.sh; gcc's libasan as the runtime),
# Python's own allocations through malloc so that ctypes buffers and numpy arrays carry red zones too, and the normal
# interpreter exit (no os._exit guard).  Reports: gpurun_out/r05_asan/asan.<pid>; the suite's output: suite.txt.
# Run on the GPU box:  bash scripts/asan_gpu_suite.sh [pytest args...]
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_asan
mkdir -p $OUT
TORCH_LIB=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))" 2>/dev/null)
export PYTHONMALLOC=malloc
# libstdc++ next to libasan: the interceptor of __cxa_throw resolves the real one when the runtime starts, and python itself
# does not link libstdc++ (without it the first C++ exception — torch throws one while probing nvrtc — ends in a CHECK failure)
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) /usr/lib/x86_64-linux-gnu/libstdc++.so.6"
# the runtime's dlopen interceptor makes libasan the caller of every dlopen: a library found through its loader's RUNPATH
# ($ORIGIN of torch/lib: libcaffe2_nvrtc.so) is no longer found without help
export LD_LIBRARY_PATH=$TORCH_LIB:$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:log_path=$PWD/$OUT/asan
export PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_hostasan.so
export PRIMME_AMD_HOSTCHECK_LIB=$PWD/oracle/_build/libprimme_hostcheck_asan.so
export PRIMME_AMD_TEST_NORMAL_EXIT=1
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
rc=$?
echo "smoke under host-ASan: exit $rc" | tee -a $OUT/smoke.txt
if [ $rc -ne 0 ]; then
   tail -30 $OUT/smoke.txt
   echo "the instrumented stack does not start"
   [ -z "$ASAN_FALLBACK_PLAIN" ] && exit 1
   unset LD_PRELOAD PRIMME_AMD_LIB PRIMME_AMD_HOSTCHECK_LIB ASAN_OPTIONS
   export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
fi
timeout ${SUITE_TIMEOUT:-1500} python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider "$@" > $OUT/suite.txt 2>&1
echo "suite exit $?" | tee -a $OUT/suite.txt
tail -15 $OUT/suite.txt
ls -la $OUT
for f in $OUT/asan.*; do [ -f "$f" ] && { echo "== $f"; grep -m3 -A12 "ERROR: AddressSanitizer" "$f" | head -60; }; done
