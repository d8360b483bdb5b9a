#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c11; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "mixer_b16 MLPK_P8_REVERSE=$v: "
    MLPK_P8_REVERSE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done
done 2>&1 | tee $OUT/ab_p8_reverse.txt
for v in 0 1; do echo -n "mixer_l16 MLPK_P8_REVERSE=$v: "; MLPK_P8_REVERSE=$v timeout 300 python bench.py --model mixer_l16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a $OUT/ab_p8_reverse.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "p8 or two_tile" 2>&1 | tail -2
