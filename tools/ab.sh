#!/bin/bash
# measurement only (GPU box, repository root): tools/ab.sh <c3|long> <rounds> <launches> — every build/lib_*.so in turn, <rounds> times
for r in $(seq 1 ${2:-3}); do
  for f in build/lib_*.so; do
    SVJG_HIP_LIB=$PWD/$f timeout -k 10 300 python tools/ab_one.py ${1:-c3} ${3:-300} 2>&1 | tail -1
  done
done
