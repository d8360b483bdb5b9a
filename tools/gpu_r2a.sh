#!/bin/bash
# round-2 GPU visit A: parity of the restructured persistent GEMM + first measurements
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2a
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest gemm first" ; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_gemm.log
echo "== calib variants"; timeout 300 python tools/gemm_p8_calib.py 20 > $OUT/calib_variants.txt 2>&1; tail -30 $OUT/calib_variants.txt
for ni in 1 2 3 4; do MLPK_P8_FORCE_NI=$ni timeout 120 python tools/gemm_p8_calib.py 20 >> $OUT/calib_ni.txt 2>&1; done; cat $OUT/calib_ni.txt
echo "== pytest all"; timeout 900 python -m pytest tests -q -m gpu -rA > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
grep -E "^(real|block) " $OUT/pytest_gpu.log > $OUT/parity_lines.txt
echo "== bench"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
