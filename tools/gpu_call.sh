#!/bin/bash
# One visit to the GPU box: tools/gpu_call.sh LABEL 'shell commands' -- runs them from the repo root with the environment every
# measurement needs, and keeps the commands next to their output (gpurun_out/LABEL/log.txt), so the record of what produced a file
# in profiles/ is the log's first lines instead of one committed script per call.
#   /usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/gpu_call.sh r5c1 'python bench.py; ...'"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONDONTWRITEBYTECODE=1
LABEL="$1"; shift
O=$PWD/gpurun_out/$LABEL; mkdir -p "$O"; export O
{ echo "# $(date -u +%FT%TZ) $LABEL"; echo "# commands: $*"; } > "$O/log.txt"
bash -c "$*" 2>&1 | tee -a "$O/log.txt"
