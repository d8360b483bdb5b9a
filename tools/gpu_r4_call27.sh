#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mixshift or shift or axial" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "msmlp or ms_mlp or asmlp or as_mlp" 2>&1 | tail -4
for m in msmlp_t asmlp_t; do
  timeout 300 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$m', d['value'], d['ms_per_step'])" | tee -a $O/bench.txt
done
