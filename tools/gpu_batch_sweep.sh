#!/bin/bash
# throughput of a model at several batch sizes in one session (does a smaller batch -- intermediate tensors inside the 256 MB Infinity Cache -- run
# faster per image?): bash tools/gpu_batch_sweep.sh model batch...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
m=$1; shift
for b in "$@"; do
  echo -n "$m batch $b: "
  timeout 300 python bench.py --model $m --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms')"
done
