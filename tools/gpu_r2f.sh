#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2f
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest all"; timeout 900 python -m pytest tests -q -m gpu -rA > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
grep -E "^(real|block) " $OUT/pytest_gpu.log > $OUT/parity_lines.txt
