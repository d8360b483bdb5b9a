#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "token_mlp" 2>&1 | tail -4
echo "== PF=3"; timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2
for v in 0 1 2; do echo "== PF=$v"; MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_pf$v.so timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2; done
echo "== PF=3"; timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2
echo "== bench"; timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
