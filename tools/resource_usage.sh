#!/bin/bash
# housekeeping (CPU, no GPU needed): what the compiler says every kernel of libsvjg_hip.so uses — VGPRs, SGPR spills, scratch, LDS,
# occupancy — from -Rpass-analysis=kernel-resource-usage of the shipped flags, so that DESIGN.md quotes the compiler and not memory.
#   bash tools/resource_usage.sh > profiles/r04/resource_usage.txt
R=$(cd "$(dirname "$0")/.." && pwd)
echo "kernel resource usage of libsvjg_hip.so (hipcc --offload-arch=gfx950 -O3, $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*'), $(date -u +%F); git $(git -C $R rev-parse --short HEAD 2>/dev/null || cut -d" " -f1 $R/tools/_build/git_head 2>/dev/null || echo unknown)$(git -C $R rev-parse --short HEAD >/dev/null 2>&1 && { git -C $R diff --quiet || echo +changes; }))"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -I/opt/rocm/include -o /tmp/svjg_ru.so $R/svjedi-graph_amd/csrc/svjg_capi.hip \
  -L/opt/rocm/lib -lrccl -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|SGPRs Spill|TotalSGPRs|Occupancy|LDS Size" \
  | sed 's/.*remark: *//; s/ *\[-Rpass[^]]*\]//' | awk '/Function Name/ { if (line) print line; cmd = "c++filt " $3; cmd | getline nm; close(cmd); line = nm ":" ; next } { gsub(/^ +/, ""); line = line "  " $0 } END { print line }'
rm -f /tmp/svjg_ru.so
