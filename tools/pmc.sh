#!/bin/bash
# measurement only: SQ counter passes of the bench command (one rocprofv3 run per counter group)
# usage: tools/pmc.sh <outdir> ; run from the repository root on the GPU box
out=${1:-gpurun_out/pmc}
mkdir -p $out
export TMPDIR=/tmp
i=0
GROUPS_DEFAULT=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC")
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -ra GROUPS_DEFAULT <<< "$PMC_GROUPS"; fi
for grp in "${GROUPS_DEFAULT[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 2 --warmup 1 > $out/g$i.log 2>&1
  f=$(ls -t $out/g$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_classify_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, sum(v) / len(v), "per launch (", len(v), "launches )")
PY
done
