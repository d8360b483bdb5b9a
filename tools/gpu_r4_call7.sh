#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c7; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "token_gemm" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_models.py -q -m gpu -k "gmlp or resmlp or gMLP or ResMLP" 2>&1 | tail -5
for m in gmlp_s resmlp_24; do
  for rep in 1 2; do
    for v in 1 0; do
      echo -n "$m MLPK_TOKEN_GEMM_LN=$v: "
      MLPK_TOKEN_GEMM_LN=$v timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
    done
  done
done 2>&1 | tee $OUT/ab_token_gemm_ln.txt
