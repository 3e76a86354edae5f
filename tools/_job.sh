cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j12
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quirks or fuzz or random_graphs or unicode or realshape or dover or deferral or identity_tag or paths_of_65 or long_paths or lines_longer or stripes or node_names" > gpurun_out/j12/tests.log 2>&1; tail -4 gpurun_out/j12/tests.log
for i in 1 2; do ALL_SLOW=1 python3 tools/slowpath_bench.py 0 2>&1 | tail -1; done | tee gpurun_out/j12/slowpath_all_slow.txt
python3 tools/slow_long_probe.py 2>&1 | tee gpurun_out/j12/slow_long_probe.txt
