#!/bin/bash
# temporal vs non-temporal stores of the generated q4 tiles, by output size (MLPK_Q4_TEMPORAL_MB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c35; mkdir -p $O
python -c "
import importlib,sys; sys.path.insert(0,'.'); p=importlib.import_module('jittor-mlp_amd'); print('lib ok', p._native.lib().mlpk_abi_version())" || exit 1
for m in gmlp_s resmlp_24 vip_s7 s2mlpv2; do
  for mb in 0 260; do
    MLPK_Q4_TEMPORAL_MB=$mb timeout 120 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$m MLPK_Q4_TEMPORAL_MB=$mb', d['value'], d['ms_per_step'])" | tee -a $O/ab_q4_temporal.txt
  done
done
