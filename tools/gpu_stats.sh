#!/bin/bash
# A/B of the by-product row statistics (GEMM epilogue -> mean / rstd) against the separate statistics pass, same box.
# Every step under its own timeout; nothing here reads stdin.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/stats
mkdir -p $OUT
echo "== op tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "row_stat or row_parts or p8 or rowmajor" > $OUT/ops.log 2>&1 < /dev/null; tail -2 $OUT/ops.log
echo "== model tests"; timeout 600 python -m pytest tests/test_gpu_models.py -q > $OUT/models.log 2>&1 < /dev/null; tail -4 $OUT/models.log
for m in vip_s7 gmlp_s s2mlpv2; do
  for v in 1 0; do
    MLPK_NO_EPILOGUE_STATS=$v timeout 120 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline < /dev/null 2>>$OUT/err.log > $OUT/b.json
    timeout 20 python -c "import json,sys; d=json.loads(open('$OUT/b.json').readline()); print('$m separate_pass=$v', d['value'], d['ms_per_step'])" < /dev/null
  done
done
timeout 200 bash tools/prof_model.sh vip_s7 < /dev/null 2>&1 | tail -12
