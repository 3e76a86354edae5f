#!/bin/bash
# device-only ISA of libsvjg_hip for gfx950 -> /tmp/isa/capi.s, k_classify_main alone -> /tmp/isa/main.s
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I/opt/rocm/include --cuda-device-only -S \
    -DSVJG_MARK "$@" -o /tmp/isa/capi.s "$ROOT/svjedi-graph_amd/csrc/svjg_capi.hip" -Rpass-analysis=kernel-resource-usage 2> /tmp/isa/remarks.txt
awk '/^_ZN4svjg15k_classify_mainENS_12ClassifyArgsE:/,/s_endpgm/' /tmp/isa/capi.s > /tmp/isa/main.s
grep -c " error" /tmp/isa/remarks.txt | sed "s/^/compile errors: /"; grep -A9 'Function Name: _ZN4svjg15k_classify_main' /tmp/isa/remarks.txt | grep -o 'remark:.*' | sed 's/ \[-Rpass.*//'
