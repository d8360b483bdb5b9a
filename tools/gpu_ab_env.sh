#!/bin/bash
# A/B of an environment switch in one session: bench.py with VAR unset vs VAR=VALUE; usage: gpu_ab_env.sh VAR VALUE [model]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
var=$1; val=$2; m=${3:-mixer_b16}
for rep in 1 2 3; do
  for v in default "$var=$val"; do
    unset $var
    [ "$v" = default ] || export $var=$val
    echo -n "$m $v: "
    timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
