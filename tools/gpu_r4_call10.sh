#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c10; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
b() { timeout 300 python bench.py --model $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "axial_shift_core or q4_generated" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "asmlp or AS_MLP or as_mlp or AxialShift" 2>&1 | tail -4
for rep in 1 2; do for v in 1 0; do echo -n "asmlp_t MLPK_ASMLP_FUSED_CONV2=$v: "; MLPK_ASMLP_FUSED_CONV2=$v b asmlp_t; done; done 2>&1 | tee $OUT/ab_asmlp_fused.txt
for v in 0 1; do echo -n "asmlp_t MLPK_ASMLP_EPILOGUE_STATS=$v: "; MLPK_ASMLP_EPILOGUE_STATS=$v b asmlp_t; done 2>&1 | tee -a $OUT/ab_asmlp_fused.txt
timeout 300 bash tools/prof_model.sh asmlp_t < /dev/null 2>&1 | tail -16
for m in s2mlpv2 vip_s7 mixer_b16; do echo -n "$m: "; b $m; done
