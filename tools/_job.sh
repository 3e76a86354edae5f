cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j7
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quirks or fuzz or random_graphs or unicode or realshape or dover or deferral or identity_tag or paths_of_65 or long_paths or lines_longer or stripes" > gpurun_out/j7/tests.log 2>&1; tail -4 gpurun_out/j7/tests.log
python3 - <<PY > gpurun_out/j7/long_read.txt 2>&1
import os, sys, json, tempfile
sys.path[:0] = [os.getcwd(), os.getcwd() + "/svjedi-graph_amd", os.getcwd() + "/tools"]
import bench, synth
from svjg import capi
from svjg.graph import Graph
ctx = capi.Context(0)
print(json.dumps(bench.long_read_block(capi, synth, Graph, ctx, tempfile.mkdtemp(), check=True)))
PY
cat gpurun_out/j7/long_read.txt
ALL_SLOW=1 python3 tools/slowpath_bench.py 0 > gpurun_out/j7/slowpath_all_slow.txt 2>&1; tail -2 gpurun_out/j7/slowpath_all_slow.txt
