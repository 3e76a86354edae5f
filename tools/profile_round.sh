#!/bin/bash
# measurement only: everything profiles/<round>/ holds, in one call on the GPU box (bench lines of the three workloads, rocprofv3
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE and three SQ counter groups in separate --pmc passes, traffic.json from them).
#   gpurun -- bash tools/profile_round.sh r03 ; then copy the summaries from gpurun_out/r03/ into profiles/r03/
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-profile}; mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 bench.py --workload c3 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench_c3.err
python3 bench.py --workload c4shard --no-cpu-baseline > $O/bench_c4shard.json 2>> $O/bench_c3.err
SVJG_BENCH_SYNC=1 python3 bench.py --workload c3 --no-cpu-baseline --no-e2e > $O/bench_c3_one_pass_at_a_time.json 2>> $O/bench_c3.err
B="python3 bench.py --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --stats -d $O/kt -o k --output-format csv -- $B --steps 10 --warmup 2 > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $O/sq1 -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq2 -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/sq3 -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/sq3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/tcc.log 2>&1 || true
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/grbm -o p --output-format csv -- $B --steps 3 --warmup 1 > $O/grbm.log 2>&1 || true
# flat names for profiles/<round>/
cp $(ls $O/kt/*kernel_stats.csv | head -1) $O/kernel_stats_bench_c3.csv
for n in fetch:pmc_fetch_size write:pmc_write_size sq1:pmc_sq_group1 sq2:pmc_sq_group2 sq3:pmc_sq_group3 tcc:pmc_tcc grbm:pmc_grbm; do
  f=$(ls $O/${n%%:*}/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -E 'Counter_Name|k_classify_main' "$f" > $O/${n##*:}.csv
done
python3 tools/mk_traffic.py $O
rm -rf $O/kt $O/fetch $O/write $O/sq1 $O/sq2 $O/sq3 $O/tcc $O/grbm
ls -la $O
