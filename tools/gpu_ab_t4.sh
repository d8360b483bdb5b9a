#!/bin/bash
# A/B in one process sequence: bench.py (Mixer-B/16, bs=256) with variants of the token kernel; usage: gpu_ab_t4.sh [variants...]
# a variant is "L1" (token_mlp_rr_kernel, layout 1) or a t4 dbg id (0 = shipped kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VARS="${@:-0 L1}"
for rep in 1 2 3; do
  for v in $VARS; do
    unset MLPK_TOKEN_MLP_LAYOUT MLPK_T4_DBG
    if [ $v = L1 ]; then export MLPK_TOKEN_MLP_LAYOUT=1; else export MLPK_T4_DBG=$v; fi
    echo -n "variant $v: "
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
