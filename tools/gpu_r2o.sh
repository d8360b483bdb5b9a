#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
for v in b3 b4 abl2; do echo "== variant $v"; MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_$v.so timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2; done
