#!/bin/bash
# measurement only: cost split of k_classify_main via the SVJG_DIAG ablation knob
# (needs a library built with -DSVJG_ABLATE: the shipped kernel does not test the knob;
#  hipcc ... -DSVJG_ABLATE -o svjedi-graph_amd/csrc/libsvjg_hip.so, see __graft_entry__.build_hip for the full command)
for d in ${DIAGS:-0 1 2 64 4 8}; do
  echo "SVJG_DIAG=$d $(SVJG_DIAG=$d python bench.py --workload ${1:-c3} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print(r["kernel_ms"], r["roofline"]["achieved"])')"
done
