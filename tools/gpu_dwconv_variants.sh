#!/bin/bash
# Builds the depthwise kernel in several variants ON the GPU box and times each on the ConvMixer-1536/20 layer shape.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/dwv; mkdir -p $O; : > $O/results.txt
for v in "$@"; do
  tag=$(echo "$v" | tr -d ' =' | tr -c 'A-Za-z0-9_\n' '_')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Ijittor-mlp_amd/csrc -Iinclude $v tools/ubench/dwconv_bench.cpp jittor-mlp_amd/csrc/mlpk_dwconv.hip -o /tmp/dw_$tag 2> $O/build_$tag.err || { echo "$v: build failed" | tee -a $O/results.txt; continue; }
  for i in 1 2; do echo -n "[$v] " | tee -a $O/results.txt; timeout 120 /tmp/dw_$tag | tee -a $O/results.txt; done
done
