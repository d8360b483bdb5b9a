#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "swin_spatial" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "swin" 2>&1 | tail -6
for f in 1 0 1 0; do
  MLPK_SWIN_SPATIAL_FUSED=$f timeout 300 python bench.py --model swinmlp_t --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('swinmlp_t MLPK_SWIN_SPATIAL_FUSED=$f', d['value'], d['ms_per_step'])" | tee -a $O/ab_swin_spatial.txt
done
