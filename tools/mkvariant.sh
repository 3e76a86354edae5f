#!/bin/bash
# measurement only: build/lib_<name>.so = libsvjg_hip.so of the working tree with extra compiler flags
#   tools/mkvariant.sh <name> [flags...]   then on the GPU box: tools/variants.sh
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -I/opt/rocm/include "$@" -o $R/build/lib_$name.so \
  $R/svjedi-graph_amd/csrc/svjg_capi.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 'Function Name: _ZN4svjg15k_classify_mainE' | grep -E "VGPRs:|Scratch|SGPRs Spill|Occupancy" | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' '; echo " <- $name"
