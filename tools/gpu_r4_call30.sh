#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "shift or axial" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "asmlp or as_mlp" 2>&1 | tail -3
