#!/bin/bash
# round 5, step 10: row-pattern SpMV in pairs of rows: resident waves 4 vs 6, in isolation and inside the two Laplacian solves
cd "$(dirname "$0")/.."
O=gpurun_out/r05_step10; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "pattern or csr or halo or matvec" > $O/kernel_tests.txt 2>&1; echo "kernel tests exit $?"; tail -2 $O/kernel_tests.txt
for v in r1_w4 r1_w6; do
  export PRIMME_AMD_LIB=$PWD/primme_amd/variants/libprimme_amd_pat_$v.so
  echo "== $v"; timeout 300 python scripts/spmv_format_perf.py 100 2>&1 | grep -v amdgpu.ids | grep "format 2\|bit\|differ" | tee $O/perf_$v.txt
  for wl in lap3d_2m lap2d_10m; do
    if [ $wl = lap3d_2m ]; then ST="--steps 3 --warmup 1"; else ST="--steps 1 --warmup 0"; fi
    timeout 900 python bench.py --workload $wl $ST --no-configs1 --no-extra-configs --no-cpu-baseline > $O/bench_${v}_$wl.json 2> $O/bench_${v}_$wl.err
    python - $O/bench_${v}_$wl.json <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric"')][0]
    c = d["config"]; k = d["roofline"]["all_kernels"]["csr_stream_kernel"]
    print("  ", c["workload"][:10], d["value"], "eig/s", c["us_per_outer_iteration"], "us/iter", c["outer_iterations"], "its; spmv class", round(1e3 * k["ms"] / k["launches"], 2), "us/launch", k.get("GBps_streamed"), "GB/s streamed")
except Exception as e:
    print("  no line:", e)
PY
  done
done
