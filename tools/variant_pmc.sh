#!/bin/bash
# measurement only (GPU box): for every build/lib_*.so variant of libsvjg_hip.so: bench line (kernel ms) and one rocprofv3 --pmc pass
# (SQ_INSTS_VALU / SALU / LDS, SQ_WAVE_CYCLES) of the same command; DIAGS="0 1 2" also walks the SVJG_DIAG ablation knob of
# -DSVJG_ABLATE builds.  Output: gpurun_out/variants/<name>.{json,pmc}
#   gpurun -- bash tools/variant_pmc.sh [workload]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/variants; mkdir -p $O
export TMPDIR=/tmp
cd $R
W=${1:-c3}

for f in build/lib_*.so; do
  n=$(basename $f .so)
  export SVJG_HIP_LIB=$R/$f
  for d in ${DIAGS:-0}; do
    export SVJG_DIAG=$d
    timeout -k 10 200 python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 10 --warmup 2 > $O/$n.d$d.json 2> $O/$n.d$d.err || { echo "$n diag $d: bench failed"; tail -3 $O/$n.d$d.err; continue; }
    ms=$(python3 -c "import json,sys; r=json.load(open('$O/$n.d$d.json')); print(round(r['kernel_ms']['classify_main'],4), r['deferred_lines_per_step'], r['genotyped_rows'])")
    line="$n diag=$d ms,deferred,genotyped= $ms"
    if [ -z "$NO_PMC" ]; then
      rm -rf $O/pmc_$n
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/pmc_$n -o p --output-format csv -- python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 2 --warmup 1 > $O/$n.d$d.pmclog 2>&1
      f2=$(ls -t $O/pmc_$n/*counter_collection.csv 2>/dev/null | head -1)
      [ -n "$f2" ] && line="$line $(python3 - "$f2" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_classify_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(" ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/1e6:.1f}M" for k, v in sorted(acc.items())))
PY
)"
      rm -rf $O/pmc_$n
    fi
    echo "$line" | tee -a $O/summary.txt
  done
done
