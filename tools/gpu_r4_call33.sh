#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "short_k_linear_gelu" 2>&1 | tail -12
for m in gmlp_s resmlp_24 vip_s7 s2mlpv2 asmlp_t; do
  for f in 1 0; do
    MLPK_LINEAR_GELU=$f timeout 300 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$m MLPK_LINEAR_GELU=$f', d['value'], d['ms_per_step'])" | tee -a $O/ab_linear_gelu.txt
  done
done
