#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2c
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest gemm"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > $OUT/pytest_gemm.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_gemm.log
echo "== A/B"; timeout 400 python tools/gemm_ab.py 7 > $OUT/gemm_ab.txt 2>&1; grep -v amdgpu.ids $OUT/gemm_ab.txt | head -24
for c in 1 2 4; do echo "== bench chunks=$c"; MLPK_CHANNEL_CHUNKS=$c timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_chunks$c.json 2>> $OUT/bench.err; python - <<PY
import json
d = json.load(open("$OUT/bench_chunks$c.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: (v["avg_ms"], v["launches"]) for k, v in d["kernels"].items()})
PY
done
echo "== models test"; timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "real or batch_256" > $OUT/pytest_models.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_models.log
