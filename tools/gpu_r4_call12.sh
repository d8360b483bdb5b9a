#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c12; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
b() { timeout 300 python bench.py --model $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "statistics or finalize" 2>&1 | tail -2
for m in mixer_b16 gmlp_s vip_s7; do for rep in 1 2; do for v in 4 1; do echo -n "$m MLPK_FINALIZE_LANES=$v: "; MLPK_FINALIZE_LANES=$v b $m; done; done; done 2>&1 | tee $OUT/ab_finalize.txt
