#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== models"; timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "sparse" 2>&1 | tail -3
for m in sparsemlp_t; do for v in "" "MLPK_NO_TOKEN_GEMM=1" ""; do env $v timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m [$v] %9.1f img/s %7.2f ms' % (d['value'], d['ms_per_step']))"; done; done
bash tools/prof_model.sh sparsemlp_t 2>&1 | head -12
