#!/bin/bash
# measurement only: bench every build/lib_*.so variant of libsvjg_hip.so (run from the repository root on the GPU box).  The variant is
# selected through SVJG_HIP_LIB (svjg/capi.py: load_library); the shipped library is never touched.
for f in build/lib_*.so; do
  export SVJG_HIP_LIB=$PWD/$f
  echo "$(basename $f) sync $(SVJG_BENCH_SYNC=1 timeout -k 10 200 python bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 12 --warmup 3 2>&1 | grep -o '"kernel_ms[^}]*}') pipe $(timeout -k 10 200 python bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 12 --warmup 3 2>&1 | grep -o '"value": [0-9.]*')"
done
