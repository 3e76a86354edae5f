cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j13
for lam in 3 5 8 12; do
  for W in c4shard c3; do
    echo "lambda=$lam $W $(SVJG_NAME_LAMBDA=$lam SVJG_BENCH_SYNC=1 python3 bench.py --workload $W --no-cpu-baseline --no-e2e --no-north-star --no-long-read --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['kernel_ms']['classify_main'],4), r['setup_s'])")"
  done
done | tee gpurun_out/j13/lambda.txt
