#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c8; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "skinny or gemm or token_gemm or statistics" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -k "gmlp or gMLP or vip or ViP or s2 or cycle or Cycle" 2>&1 | tail -5
for m in vip_s7 s2mlpv2 cyclemlp_b1; do
  for rep in 1 2; do
    for v in 1 0; do
      echo -n "$m MLPK_GEMM_SKINNY=$v: "
      MLPK_GEMM_SKINNY=$v timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    done
  done
done 2>&1 | tee $OUT/ab_skinny.txt
for rep in 1 2; do
  for v in 1 0; do
    echo -n "gmlp_s MLPK_TOKEN_GEMM_LN=$v: "
    MLPK_TOKEN_GEMM_LN=$v timeout 300 python bench.py --model gmlp_s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee $OUT/ab_gmlp.txt
for m in gmlp_s resmlp_24 vip_s7; do timeout 300 bash tools/prof_model.sh $m < /dev/null 2>&1 | tail -14; done
