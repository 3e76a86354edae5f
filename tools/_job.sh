cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j14
SVJG_HIP_LIB=$PWD/build/lib_pfdisp.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quirks or synth_g6 or random_graphs or fuzz or node_names or paths_of_65 or realshape or c2_full or stripe_boundaries or identity_tag or deferral" > gpurun_out/j14/tests_pfdisp.log 2>&1; tail -4 gpurun_out/j14/tests_pfdisp.log
for W in c3 c4shard; do
  for i in 1 2; do
    echo "$W shipped $(SVJG_BENCH_SYNC=1 python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],4))")"
    echo "$W pfdisp  $(SVJG_HIP_LIB=$PWD/build/lib_pfdisp.so SVJG_BENCH_SYNC=1 python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],4))")"
  done
done | tee gpurun_out/j14/pfdisp.txt
