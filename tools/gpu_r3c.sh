#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "layernorm_transpose" 2>&1 | tail -3
echo "== models"; timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "gmlp" 2>&1 | tail -3
timeout 300 python bench.py --model gmlp_s --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gmlp_s %9.1f img/s %7.2f ms' % (d['value'], d['ms_per_step']))"
bash tools/prof_model.sh gmlp_s 2>&1 | head -6
