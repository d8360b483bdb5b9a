#!/bin/bash
# dwconv: prefetch loads without the per-load vmcnt(0), taps through the LDS.  Tests, ConvMixer bench, counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4c20; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dwconv or convmixer or conv_mixer" 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --model convmixer_1536_20 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('convmixer_1536_20', d['value'], d['ms_per_step'], json.dumps(d.get('kernels')))" | tee -a $O/bench.txt; done
bash tools/pmc_model.sh convmixer_1536_20 dwconv > $O/pmc.log 2>&1; tail -25 $O/pmc.log
