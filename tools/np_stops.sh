#!/bin/bash
# measurement only (GPU box): VALU instructions and time of k_classify_main when the node pass ends after its k-th part
# (build/lib_np<k>.so = -DSVJG_NP_STOP=k builds; lib_np9.so = the shipped kernel).  Output: gpurun_out/np_stops.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
cp svjedi-graph_amd/csrc/libsvjg_hip.so /tmp/lib_keep.so
for f in build/lib_np*.so; do
  n=$(basename $f .so); cp $f svjedi-graph_amd/csrc/libsvjg_hip.so
  rm -rf /tmp/pmc_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d /tmp/pmc_$n -o p --output-format csv -- python3 bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --steps 2 --warmup 1 > /tmp/$n.log 2>&1
  ms=$(timeout -k 10 100 python3 bench.py --workload ${1:-c3} --no-cpu-baseline --no-e2e --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],4))")
  f2=$(ls -t /tmp/pmc_$n/*counter_collection.csv 2>/dev/null | head -1)
  echo "$n ms=$ms $(python3 - "$f2" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_classify_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(" ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/1e6:.1f}M" for k, v in sorted(acc.items())))
PY
)" | tee -a $O/np_stops.txt
done
cp /tmp/lib_keep.so svjedi-graph_amd/csrc/libsvjg_hip.so
