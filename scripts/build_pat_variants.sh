#!/bin/bash
# A/B builds of the row-pattern SpMV kernel only (rows per lane and trip, resident waves per SIMD): the other objects are the default build's.
#   bash scripts/build_pat_variants.sh "2 8" "8 4" ...   ->  primme_amd/variants/libprimme_amd_pat_r<RPL>_w<WPS>.so
set -e
cd "$(dirname "$0")/../primme_amd/csrc"
make -s all
mkdir -p ../variants
for v in "$@"; do
  set -- $v; R=$1; W=$2; X=${3:-}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DHIPK_PAT_RPL=$R -DHIPK_PAT_WPS=$W $X -c hipk_sparse_pat.hip -o /tmp/pat_r${R}_w${W}.o
  OBJS=$(ls *.o | grep -v '^hipk_sparse_pat.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,--version-script=exports.map -o ../variants/libprimme_amd_pat_r${R}_w${W}.so \
     $OBJS /tmp/pat_r${R}_w${W}.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -lm
  ls -la ../variants/libprimme_amd_pat_r${R}_w${W}.so
done
