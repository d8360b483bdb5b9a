#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c6; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -1 $OUT/build.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_a.json
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "statistics or row_parts or q4" 2>&1 | tail -4
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_b.json
