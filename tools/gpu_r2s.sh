#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2s; mkdir -p $OUT
python __graft_entry__.py build > /dev/null 2>&1
echo "== split test"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "column_split or p8" 2>&1 | tail -4
for m in vip_s7 resmlp_24 s2mlpv2 gmlp_s cyclemlp_b1 hiremlp_s; do
  for v in "" "MLPK_GEMM_NO_NSPLIT=1"; do
    env $v timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-24s %9.1f img/s %7.2f ms' % ('$m', '$v', d['value'], d['ms_per_step']))"
  done
done
