#!/bin/bash
# Build-time variants of the product library for A/B measurements (never what the tests or the bench load by default):
#   bash scripts/build_variant.sh <name> "<extra hipcc flags>"   ->  primme_amd/variants/libprimme_amd_<name>.so
# e.g.  bash scripts/build_variant.sh nt1 "-DHIPK_NT_LOADS=1";  PRIMME_AMD_LIB=primme_amd/variants/libprimme_amd_nt1.so python bench.py
set -e
cd "$(dirname "$0")/../primme_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p ../variants/obj_$NAME
make -s all
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. $FLAGS -c $f -o ../variants/obj_$NAME/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,--version-script=exports.map -o ../variants/libprimme_amd_$NAME.so \
   ../variants/obj_$NAME/*.o $(ls *.c | sed 's/\.c$/.o/') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lm
rm -rf ../variants/obj_$NAME
ls -la ../variants/libprimme_amd_$NAME.so
