#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
for m in gmlp_s resmlp_24 s2mlpv2 hiremlp_s sparsemlp_t cyclemlp_b1 mixer_s16 vip_s7; do
  for v in new old new old; do
    if [ $v = old ]; then export MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_oldcost.so; else unset MLPK_LIB_PATH; fi
    timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-4s %9.1f img/s %7.2f ms' % ('$m', '$v', d['value'], d['ms_per_step']))"
  done
done
