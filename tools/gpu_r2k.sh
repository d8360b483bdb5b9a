#!/bin/bash
# visit K: token-MLP epilogue without per-pass vmcnt(0); scalar-vs-packed GELU A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2k
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "token_mlp or layernorm_transpose" 2>&1 | tail -3
echo "== default build"; timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -3
for f in "-DTM_GELU_SCALAR"; do
  echo "== build $f"; MLPK_EXTRA_FLAGS="$f" python __graft_entry__.py build > $OUT/build2.log 2>&1
  MLPK_EXTRA_FLAGS="$f" timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -3
done
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== models"; timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "mixer" 2>&1 | tail -3
echo "== bench"; timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
