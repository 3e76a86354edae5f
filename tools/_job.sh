cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j8
python -m pytest tests -x -q -m gpu > gpurun_out/j8/tests_all.log 2>&1; tail -5 gpurun_out/j8/tests_all.log
ALL_SLOW=1 python3 tools/slowpath_bench.py 0 > gpurun_out/j8/slowpath_all_slow.txt 2>&1; tail -2 gpurun_out/j8/slowpath_all_slow.txt
python3 bench.py --no-north-star > gpurun_out/j8/bench_c3.json 2> gpurun_out/j8/bench_c3.err; python3 -c "
import json; r=json.load(open('gpurun_out/j8/bench_c3.json')); print(r['value']/1e9, r['ms_per_step'], r['kernel_ms'], r['roofline']); print(r['long_read'])"
bash tests/fuzz_campaign.sh 20000 800 20 > gpurun_out/j8/fuzz.log 2>&1; tail -3 gpurun_out/j8/fuzz.log; cp gpurun_out/fuzz/campaign.txt gpurun_out/j8/fuzz_campaign.txt
