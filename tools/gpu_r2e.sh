#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2e
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest all"; timeout 900 python -m pytest tests -q -m gpu -rA > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
grep -E "^(real|block) " $OUT/pytest_gpu.log > $OUT/parity_lines.txt
: > $OUT/bench_models.jsonl
for m in asmlp_t cyclemlp_b1 vip_s7 s2mlpv2 gmlp_s resmlp_24 convmixer_1536_20; do timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r2e/bench_models.jsonl"):
    d = json.loads(l)
    print("%-40s %10.1f img/s %8.2f ms  %7.1f model-TF/s" % (d["metric"], d["value"], d["ms_per_step"], d["model_tflops"]))
PY
bash tools/prof_model.sh asmlp_t 2>&1 | tail -16
