for l in 3 4 5 6 8; do
echo "lambda=$l $(SVJG_NAME_LAMBDA=$l SVJG_VERBOSE=1 SVJG_BENCH_SYNC=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --steps 12 --warmup 3 2>/tmp/err.txt | grep -o '"kernel_ms[^}]*}') $(grep -o 'hash and displace: [0-9.]* s' /tmp/err.txt | head -1)"
done
