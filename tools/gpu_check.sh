#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: tools/gpu_check.sh [tests|bench|prof|all]...   (default: all)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
WHAT="${*:-all}"
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/rocminfo.txt
nproc >> $OUT/rocminfo.txt
if has tests; then
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -rA --durations=15 > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
fi
if has bench; then
  timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.json
fi
if has models; then
  : > $OUT/bench_models.jsonl
  for m in mixer_s16 mixer_l16 gmlp_s resmlp_24 vip_s7 s2mlpv2 asmlp_t convmixer_1536_20 sparsemlp_t hiremlp_s msmlp_t swinmlp_t; do
    timeout 600 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err
  done
  python - <<'PY'
import json
for l in open("gpurun_out/bench_models.jsonl"):
    d = json.loads(l)
    print("%-60s %10.1f img/s %8.2f ms  %7.1f model-TF/s" % (d["metric"], d["value"], d["ms_per_step"], d["model_tflops"]))
PY
fi
if has prof; then
  rm -rf $OUT/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o mixer_b16 -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-variants > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" )
  echo "prof rc=$?"
  find $OUT/prof -name "*kernel_stats*" | head -3
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f"
  # keep only the small summaries (the raw kernel trace can be large)
  find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
fi
