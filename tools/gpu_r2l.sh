#!/bin/bash
# visit L: token-MLP ablations (timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
echo "== production"; timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2
for a in 1 2 4 8 3 12 15; do
  echo "== ABL=$a"; MLPK_LIB_PATH=$PWD/jittor-mlp_amd/lib/variants/libmlpk_abl$a.so timeout 300 python tools/tokenmlp_timeline.py 2>&1 | tail -2
done
