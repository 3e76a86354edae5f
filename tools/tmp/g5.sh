for d in 1 4 16 64; do
echo "hot=$d $(SVJG_SYNTH_HOT=$d SVJG_BENCH_SYNC=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --steps 12 --warmup 3 2>&1 | grep -o '"kernel_ms[^}]*}')"
done
