#!/bin/bash
# Same-box A/B of the generated q4 GEMM tile's automatic selection (MLPK_GEMM_Q4=0 = the round-2 tile choice) on the model benches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  env $2 timeout 150 python bench.py --model $1 --steps 20 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null > /tmp/ab.json
  timeout 20 python -c "import json; d=json.loads(open('/tmp/ab.json').readline()); print('%-18s %-22s %9.1f img/s %8.3f ms' % ('$1', '$2', d['value'], d['ms_per_step']))" < /dev/null
}
for m in ${1:-mixer_b16 mixer_l16 mixer_s16 vip_s7 gmlp_s resmlp_24 s2mlpv2 cyclemlp_b1 hiremlp_s sparsemlp_t swinmlp_t}; do
  run $m MLPK_GEMM_Q4=0
  run $m MLPK_GEMM_Q4=1
done
