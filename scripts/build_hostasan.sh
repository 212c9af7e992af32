#!/bin/bash
# Host-only AddressSanitizer build of the product library, for the heap-corruption hunt:
#   primme_amd/variants/libprimme_amd_hostasan.so
# The HOST side of every .hip file and every .c file is instrumented; the device code is not (-fno-gpu-sanitize), and the
# runtime is gcc's libasan, NOT the ROCm clang runtime: that one carries interceptors for hsa_amd_memory_pool_allocate & co.
# which expect the instrumented ROCr build and abort before the first kernel on this stack (profiles/r04_gpu_suite_exit_crash.md).
# clang's instrumentation only needs the v8 ASan ABI, which gcc 11's libasan exports.  Run with
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) PRIMME_AMD_LIB=primme_amd/variants/libprimme_amd_hostasan.so \
#   ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:log_path=gpurun_out/asan python -m pytest tests -m gpu
set -e
cd "$(dirname "$0")/../primme_amd/csrc"
OBJ=../variants/obj_hostasan
mkdir -p $OBJ
SAN="-fsanitize=address -fsanitize-recover=address -fno-omit-frame-pointer -g"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -I../../include -I. $SAN -fno-gpu-sanitize -w -c $f -o $OBJ/${f%.hip}.o &
done
for f in *.c; do
  gcc -O1 -std=c99 -fPIC -I../../include -I. -D_POSIX_C_SOURCE=200809L $SAN -w -c $f -o $OBJ/${f%.c}.o &
done
wait
g++ -shared -fPIC -fsanitize=address -Wl,-Bsymbolic -Wl,--version-script=exports.map -o ../variants/libprimme_amd_hostasan.so $OBJ/*.o \
   -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,/opt/rocm/lib -lm
rm -rf $OBJ
ls -la ../variants/libprimme_amd_hostasan.so
