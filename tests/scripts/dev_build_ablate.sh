#!/bin/bash
# development: rebuild libsvnicp_hip.so with the search kernel's ablation variants compiled in
set -e
cd "$(dirname "$0")/../../svn-icp_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DSVNICP_DEV_ABLATE -Rpass-analysis=kernel-resource-usage -c stein_split.hip -o stein_split.o > /tmp/dev_ablate.log 2>&1 || { grep error /tmp/dev_ablate.log; exit 1; }
grep -A12 "k_stein_search_bf16ILi64ELi2ELi6ELb1ELi0E" /tmp/dev_ablate.log | grep -E " VGPRs:|ScratchSize|Occupancy|LDS" | sed 's/.*remark: *//' | tr '\n' ' '; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvnicp_hip.so api.o knn_topk.o knn_scan.o knn_tiles.o spatial_prep.o stein_iter.o stein_mfma.o stein_split.o particle_update.o
