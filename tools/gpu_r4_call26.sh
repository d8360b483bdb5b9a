#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c26; mkdir -p $O
for d in 2 4 6; do echo "== MLPK_CM_DEPTH=$d" | tee -a $O/chanmlp_depth.txt; MLPK_CM_DEPTH=$d timeout 600 python tools/chanmlp_ab.py 2>&1 | grep fused | tee -a $O/chanmlp_depth.txt; done
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "channel_mlp_of_a_narrow" 2>&1 | tail -4
MLPK_CM_DEPTH=2 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "channel_mlp_of_a_narrow" 2>&1 | tail -2
MLPK_CM_DEPTH=6 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "channel_mlp_of_a_narrow" 2>&1 | tail -2
