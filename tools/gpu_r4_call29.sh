#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "shift or axial" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "asmlp or as_mlp" 2>&1 | tail -4
for f in 1 0 1 0; do
  MLPK_NORM_SHIFT_IMG=$f timeout 300 python bench.py --model asmlp_t --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('asmlp_t MLPK_NORM_SHIFT_IMG=$f', d['value'], d['ms_per_step'])" | tee -a $O/ab_norm_shift_img.txt
done
