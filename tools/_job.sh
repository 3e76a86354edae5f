cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j10
t0=$(date +%s.%N)
python3 bench.py > gpurun_out/j10/bench_c3.json 2> gpurun_out/j10/bench_c3.err
t1=$(date +%s.%N)
echo "bench.py default run: $(echo "$t1 - $t0" | bc) s wall"
python3 -c "
import json; r=json.load(open('gpurun_out/j10/bench_c3.json')); print(r['value']/1e9, r['ms_per_step'], r['kernel_ms'], r['long_read']['kernel_ms'], r['long_read']['lines_per_s']/1e9, r['north_star']['ms_per_pass'], r['e2e']['total_s'], r['cpu_baseline']['parity_on_sample'])"
python3 __graft_entry__.py smoke
