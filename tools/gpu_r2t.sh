#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "vip" 2>&1 | tail -4
echo "== models"; timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "vip" 2>&1 | tail -4
timeout 300 python bench.py --model vip_s7 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vip_s7 %9.1f img/s %7.2f ms' % (d['value'], d['ms_per_step']))"
bash tools/prof_model.sh vip_s7 2>&1 | tail -14
