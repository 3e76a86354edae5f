#!/bin/bash
# measurement only (GPU box): tests/fuzz_big.py for a range of seeds, four at a time; summary -> gpurun_out/fuzz/campaign.txt
#   gpurun -- bash tests/fuzz_campaign.sh [n_mutants] [first_seed] [n_seeds]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/fuzz; mkdir -p $O
N=${1:-20000}; S0=${2:-200}; NS=${3:-20}
: > $O/campaign.txt
for ((b = 0; b < NS; b += 4)); do
  for ((k = b; k < b + 4 && k < NS; ++k)); do
    s=$((S0 + k))
    ( timeout -k 10 900 python3 tests/fuzz_big.py $N $s > $O/seed_$s.txt 2>&1; echo "seed $s rc=$? $(tr '\n' ' ' < $O/seed_$s.txt | cut -c1-400)" >> $O/campaign.txt ) &
  done
  wait
  echo "batch $b done"; tail -4 $O/campaign.txt
done
grep -c "rc=0" $O/campaign.txt
