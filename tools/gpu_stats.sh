#!/bin/bash
# Same-box A/B of the by-product row statistics (GEMM / token-kernel epilogue -> mlpk_stats_finalize_planar) against the separate
# mlpk_row_stats pass (MLPK_NO_EPILOGUE_STATS=1), of the side-stream chains of ViP / Hire-MLP (MLPK_NO_SIDE_STREAM=1 = one stream), and of
# AS-MLP's opt-in per-sample variant.  Every step under its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # run <model> <env assignment> <label>
  env $2 timeout 120 python bench.py --model $1 --steps 20 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null > /tmp/ab.json
  timeout 20 python -c "import json; d=json.loads(open('/tmp/ab.json').readline()); print('%-14s %-34s %9.1f img/s %8.3f ms' % ('$1', '$3', d['value'], d['ms_per_step']))" < /dev/null
}
for m in ${1:-vip_s7 s2mlpv2 gmlp_s sparsemlp_t}; do
  run $m MLPK_NO_EPILOGUE_STATS=1 "separate statistics pass"
  run $m MLPK_NO_EPILOGUE_STATS=0 "statistics from epilogues"
done
for m in vip_s7 hiremlp_s; do
  run $m MLPK_NO_SIDE_STREAM=1 "one stream"
  run $m MLPK_NO_SIDE_STREAM=0 "independent chain on side stream"
done
run asmlp_t MLPK_ASMLP_EPILOGUE_STATS=0 "separate statistics pass"
run asmlp_t MLPK_ASMLP_EPILOGUE_STATS=1 "statistics from epilogues"
echo "asmlp model tests with MLPK_ASMLP_EPILOGUE_STATS=1:"
MLPK_ASMLP_EPILOGUE_STATS=1 timeout 200 python -m pytest tests/test_gpu_models.py -q -k "asmlp" < /dev/null 2>&1 | tail -1
