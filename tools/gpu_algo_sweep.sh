#!/bin/bash
# GEMM tile override sweep for tagged calls of one model: bash tools/gpu_algo_sweep.sh model "tag1 tag2" "algo1 algo2 ..."
cd "${GRAFT_REPO_ROOT:-/root/repo}"
m=$1; tags=$2; algos=$3
run() { timeout 120 python bench.py --model $m --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-variants $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
echo "$m default: $(run)"
for t in $tags; do for a in $algos; do echo "$m $t=$a: $(run "--algo $t=$a")"; done; done
echo "$m default: $(run)"
