#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "channel_mlp_of_a_narrow" 2>&1 | tail -15
timeout 600 python tools/chanmlp_ab.py 2>&1 | tee $O/chanmlp_ab.txt | tail -12
