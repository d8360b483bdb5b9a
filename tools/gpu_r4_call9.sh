#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c9; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
b() { timeout 300 python bench.py --model $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "resmlp or ResMLP or gmlp or gMLP" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "skinny" 2>&1 | tail -3
for rep in 1 2; do for v in split full rowstats; do echo -n "gmlp_s MLPK_GMLP_P1=$v: "; MLPK_GMLP_P1=$v b gmlp_s; done; done 2>&1 | tee $OUT/ab_gmlp_p1.txt
for rep in 1 2; do for v in 1 0; do echo -n "resmlp_24 MLPK_RESMLP_FOLD_G2=$v: "; MLPK_RESMLP_FOLD_G2=$v b resmlp_24; done; done 2>&1 | tee $OUT/ab_resmlp_fold.txt
for m in resmlp_24 vip_s7 s2mlpv2 mixer_s16 mixer_l16 gmlp_s hiremlp_s swinmlp_t cyclemlp_b1 sparsemlp_t asmlp_t; do
  for v in 1 2; do echo -n "$m MLPK_GEMM_Q4=$v: "; MLPK_GEMM_Q4=$v b $m; done
done 2>&1 | tee $OUT/ab_q4_everywhere.txt
