#!/bin/bash
# measurement only: everything profiles/<round>/ holds, in one call on the GPU box, PER WORKLOAD (default: c3 and c4shard = the
# north_star shard): the bench line, rocprofv3 kernel trace + stats of the same command, FETCH_SIZE / WRITE_SIZE, three SQ counter
# groups, TCC and GRBM in separate --pmc passes, and traffic.json (one entry per workload) from them.
#   python -c "import bench; bench.write_git_head()"   (the GPU box has no .git: tools/_build/git_head names the commit — with a digest of the sources — in every output)
#   gpurun -- bash tools/profile_round.sh r04 [workloads...] ; then copy the summaries from gpurun_out/r04/ into profiles/r04/
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-profile}; mkdir -p $O
shift || true
WL="${*:-c3 c4shard}"
export TMPDIR=/tmp
cd $R
python3 bench.py --workload c3 > $O/bench_c3.json 2> $O/bench.err
python3 bench.py --workload c2 --no-cpu-baseline --no-north-star --no-long-read --no-hg002-shape > $O/bench_c2.json 2>> $O/bench.err
python3 bench.py --workload c4shard --no-cpu-baseline --no-north-star --no-long-read --no-hg002-shape > $O/bench_c4shard.json 2>> $O/bench.err
SVJG_BENCH_SYNC=1 python3 bench.py --workload c3 --no-cpu-baseline --no-e2e --no-north-star --no-long-read --no-hg002-shape > $O/bench_c3_one_pass_at_a_time.json 2>> $O/bench.err
for W in $WL; do
  # (a workload of bench.py, or one of its untimed blocks alone — tools/block_run.py long_read | hg002_shape — so that no other launch of the kernel sits under its name)
  B="python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --no-hg002-shape --steps 10 --warmup 2"
  case $W in long_read|hg002_shape) B="python3 tools/block_run.py $W 6";; esac
  P=$O/raw_$W; mkdir -p $P
  echo "== $W: kernel trace" 
  rocprofv3 --kernel-trace --stats -d $P/kt -o k --output-format csv -- $B > $P/kt.log 2>&1
  cp $(ls $P/kt/*kernel_stats.csv | head -1) $O/kernel_stats_bench_$W.csv
  grep '^{' $P/kt.log | tail -1 > $O/bench_${W}_under_kernel_trace.json
  case $W in long_read|hg002_shape) cp $O/bench_${W}_under_kernel_trace.json $O/bench_$W.json;; esac
  i=0
  for grp in "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE" "sq_group1:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "sq_group2:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "sq_group3:SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
             "tcc:TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "tcc2:TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum TCC_ATOMIC_sum" \
             "ta:TA_BUSY_avr TCC_BUSY_avr" "grbm:GRBM_GUI_ACTIVE"; do
    n=${grp%%:*}; c=${grp#*:}
    echo "== $W: pmc $n"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $P/$n -o p --output-format csv -- ${B/--steps 10 --warmup 2/--steps 3 --warmup 1} > $P/$n.log 2>&1 || { echo "pass $n failed"; continue; }
    f=$(ls $P/$n/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && grep -E 'Counter_Name|k_classify_main' "$f" > $O/pmc_${n}_$W.csv
  done
  rm -rf $P
done
bash tools/resource_usage.sh > $O/resource_usage.txt 2>&1 || true
echo "commit $(cut -d" " -f1 tools/_build/git_head 2>/dev/null || echo unknown), $(date -u +%FT%TZ): every file of this directory was taken with the library built from it (tools/profile_round.sh $*)" > $O/COMMIT.txt
python3 tools/mk_traffic.py $O $WL
ls -la $O
