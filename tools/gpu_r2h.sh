#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2h
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== pytest dwconv/convmixer"; timeout 600 python -m pytest tests -x -q -m gpu -k "dwconv or convmixer" > $OUT/pytest_sub.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_sub.log
: > $OUT/bench_models.jsonl
timeout 300 python bench.py --model convmixer_1536_20 --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err
python - <<'PY'
import json
for l in open("gpurun_out/r2h/bench_models.jsonl"):
    d = json.loads(l)
    print("%-40s %10.1f img/s %8.2f ms  %7.1f model-TF/s" % (d["metric"], d["value"], d["ms_per_step"], d["model_tflops"]))
PY
bash tools/prof_model.sh convmixer_1536_20 2>&1 | head -4
