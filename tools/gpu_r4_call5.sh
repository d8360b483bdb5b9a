#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c5; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 300 python tools/q4_gelu_debug.py 2>&1 | grep -c "differ 0 of" 
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parallel.py tests/test_gpu_engine.py -q -m gpu 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_models.py -q -m gpu -s -k "batch_256" 2>&1 | grep -E "^bs256|passed|failed|Error|assert" | tee $OUT/bs256_parity.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench.json
