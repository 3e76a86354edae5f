#!/bin/bash
# measurement only (GPU box, -DSVJG_ABLATE build in build/lib_*ablate.so): kernel time against the number of workers per CU
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/occ

for f in build/lib_*ablate*.so; do
  export SVJG_HIP_LIB=$R/$f
  for o in ${OCCS:-4 5 6 7 8 9 10 12}; do
    r=$(SVJG_OCC=$o timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 6 --warmup 2 2> gpurun_out/occ/err.txt | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],3))")
    echo "$(basename $f) occ=$o ms=$r $(grep -m1 'occupancy API' gpurun_out/occ/err.txt)"
  done
done
