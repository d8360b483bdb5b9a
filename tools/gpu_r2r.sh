#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2r; mkdir -p $OUT
python __graft_entry__.py build > /dev/null 2>&1
echo "== full gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "== bench"; timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tee $OUT/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
bash tools/prof_model.sh mixer_b16 2>&1 | tail -14
