#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "channel_mlp_of_a_narrow" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -8
for m in asmlp_t swinmlp_t hiremlp_s cyclemlp_b1 sparsemlp_t s2mlpv2 msmlp_t; do
  for f in 1 0; do
    MLPK_CHANNEL_MLP_FUSED=$f timeout 300 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$m fused=$f', d['value'], d['ms_per_step'])" | tee -a $O/ab_chanmlp_models.txt
  done
done
