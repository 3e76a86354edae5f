set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 && bash tests/fuzz_campaign.sh 20000 500 20 2>&1 | tail -3
