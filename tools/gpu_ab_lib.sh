#!/bin/bash
# A/B of two builds in one session: the tree's library vs jittor-mlp_amd/lib/variants/libmlpk_<tag>.so (MLPK_LIB_PATH); usage: gpu_ab_lib.sh <tag> [models...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
MODELS="${@:-mixer_b16}"
for m in $MODELS; do
  for rep in $(seq 1 ${REPS:-3}); do
    for v in new $tag; do
      unset MLPK_LIB_PATH
      [ $v = new ] || export MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_$tag.so
      echo -n "$m $v: "
      timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    done
  done
done
