#!/bin/bash
# measurement only: bench every build/lib_*.so variant of libsvjg_hip.so (run from the repository root on the GPU box)
cp svjedi-graph_amd/csrc/libsvjg_hip.so /tmp/keep.so
for f in build/lib_*.so; do
  cp $f svjedi-graph_amd/csrc/libsvjg_hip.so
  echo "$(basename $f) sync $(SVJG_BENCH_SYNC=1 timeout -k 10 200 python bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --steps 12 --warmup 3 2>&1 | grep -o '"kernel_ms[^}]*}') pipe $(timeout -k 10 200 python bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --steps 12 --warmup 3 2>&1 | grep -o '"value": [0-9.]*')"
done
cp /tmp/keep.so svjedi-graph_amd/csrc/libsvjg_hip.so
