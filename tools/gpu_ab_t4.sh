#!/bin/bash
# A/B in one process sequence: bench.py (Mixer-B/16, bs=256) with the generated token kernel (layout 2) vs token_mlp_rr_kernel (layout 1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for lay in 2 1; do
    if [ $lay = 1 ]; then export MLPK_TOKEN_MLP_LAYOUT=1; else unset MLPK_TOKEN_MLP_LAYOUT; fi
    echo -n "layout $lay: "
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
