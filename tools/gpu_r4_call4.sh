#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c4; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 300 python tools/q4_gelu_debug.py 2>&1 | tee $OUT/q4_gelu_debug.txt
