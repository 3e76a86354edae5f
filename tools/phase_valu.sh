R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export TMPDIR=/tmp
export SVJG_HIP_LIB=$R/build/lib_ablate.so     # (selected through svjg/capi.py, never copied over the shipped library)
for d in 32 1 2 8 0; do
  export SVJG_DIAG=$d SVJG_BENCH_SYNC=1
  rm -rf /tmp/pm; timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 2 --warmup 1 > /tmp/pm.log 2>&1
  ms=$(timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],4))")
  f2=$(ls -t /tmp/pm/*counter_collection.csv 2>/dev/null | head -1)
  echo "diag=$d ms=$ms $(python3 - "$f2" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_classify_main" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(" ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/1e6:.1f}M" for k, v in sorted(acc.items())))
PY
)"
done
