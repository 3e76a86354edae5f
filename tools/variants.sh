#!/bin/bash
# measurement only: bench every build/lib_*.so variant of libsvjg_hip.so (run from the repository root on the GPU box)
for f in build/lib_*.so; do
  cp $f svjedi-graph_amd/csrc/libsvjg_hip.so
  echo "$(basename $f) $(timeout 200 python bench.py --workload ${1:-c3} --no-cpu-baseline --steps 5 --warmup 2 2>&1 | grep -o '"kernel_ms[^}]*}')"
done
