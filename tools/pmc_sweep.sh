#!/bin/bash
# measurement only (GPU box): several rocprofv3 --pmc passes over the bench command, one line of k_classify_main averages per pass.
#   gpurun -- bash tools/pmc_sweep.sh [workload]      PASSES="A B C;D E" overrides the counter groups
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc; mkdir -p $O; cd $R
export TMPDIR=/tmp
W=${1:-c3}
IFS=';' read -ra GROUPS_ <<< "${PASSES:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC;SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM;TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum;TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_BUSY_avr TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum}"
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $g -d $O/p$i -o p --output-format csv -- python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 2 --warmup 1 > $O/p$i.log 2>&1 || { echo "pass $i ($g): failed: $(grep -m1 -i 'error\|invalid\|not found' $O/p$i.log)"; continue; }
  f=$(ls -t $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_classify_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join(f"{k}={sum(v)/len(v)/1e6:.2f}M" for k, v in sorted(acc.items())))
PY
done
