#!/bin/bash
# visit M: 256-row register-resident token-MLP kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== ops"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "token_mlp" 2>&1 | tail -8
echo "== layout 1"; timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -3
echo "== layout 0"; MLPK_TOKEN_MLP_LAYOUT=0 timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2
