#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2g
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest ops+models subset first"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv or vip or cycle" > $OUT/pytest_sub.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_sub.log
echo "== pytest all"; timeout 900 python -m pytest tests -q -m gpu -rA > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
grep -E "^(real|block) " $OUT/pytest_gpu.log > $OUT/parity_lines.txt
: > $OUT/bench_models.jsonl
for m in vip_s7 convmixer_1536_20; do timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err; done
MLPK_DWCONV_NO_MFMA=1 timeout 300 python bench.py --model convmixer_1536_20 --steps 10 --warmup 3 --no-cpu-baseline >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err
python - <<'PY'
import json
for l in open("gpurun_out/r2g/bench_models.jsonl"):
    d = json.loads(l)
    print("%-40s %10.1f img/s %8.2f ms  %7.1f model-TF/s" % (d["metric"], d["value"], d["ms_per_step"], d["model_tflops"]))
PY
bash tools/prof_model.sh convmixer_1536_20 2>&1 | tail -8
bash tools/prof_model.sh vip_s7 2>&1 | tail -14
