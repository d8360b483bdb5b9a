#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c2; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
timeout 600 python tools/q4_static_probe.py all > $OUT/q4_static_probe.txt 2>&1; grep -v "bit-equal to s3: True" $OUT/q4_static_probe.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "q4 or gemm" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
