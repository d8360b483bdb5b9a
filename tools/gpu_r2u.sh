#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2u; mkdir -p $OUT
python __graft_entry__.py build > /dev/null 2>&1
timeout 600 python tools/gemm_sweep.py bf16 0 0 k128,k192,k128_big,k256_small,k320,gmlp_proj1,resmlp_fc1,vip_branch,vip_fc1,vip_fc2,s2_fc1,s2_fc2,s2_mlp1,s2_mlp2 2>&1 | grep -v amdgpu.ids
: > $OUT/bench_models.jsonl
for m in mixer_s16 gmlp_s resmlp_24 vip_s7 s2mlpv2 asmlp_t sparsemlp_t hiremlp_s msmlp_t swinmlp_t cyclemlp_b1; do timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err; done
python - <<'PY'
import json
for l in open("gpurun_out/r2u/bench_models.jsonl"):
    d = json.loads(l)
    print("%-40s %10.1f img/s %8.2f ms" % (d["metric"], d["value"], d["ms_per_step"]))
PY
