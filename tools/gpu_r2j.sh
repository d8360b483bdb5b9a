#!/bin/bash
# visit J: fused token LayerNorm (statistics+apply+transpose) and epilogue statistics of the token kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2j
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== ops"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "token_mlp or layernorm_transpose" 2>&1 | tail -5
echo "== models"; timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "mixer" 2>&1 | tail -5
for v in "" "MLPK_NO_FUSED_TOKEN_LN=1" "MLPK_NO_EPILOGUE_STATS=1" "MLPK_NO_FUSED_TOKEN_LN=1 MLPK_NO_EPILOGUE_STATS=1" ""; do
  echo "== bench [$v]"; env $v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
bash tools/prof_model.sh mixer_b16 2>&1 | tail -12
