#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c3; mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== A/B: new (static q4 + 2 waits/step + logistic GELU) vs r4b (same, polynomial GELU) vs r4a (static q4 only)"
for rep in 1 2 3; do
  for v in new r4b r4a; do
    unset MLPK_LIB_PATH
    [ $v = new ] || export MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_$v.so
    echo -n "mixer_b16 $v: "
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  done
done 2>&1 | tee $OUT/ab.txt
unset MLPK_LIB_PATH
timeout 300 python tools/q4_static_probe.py time > $OUT/q4_static_probe.txt 2>&1; cat $OUT/q4_static_probe.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
