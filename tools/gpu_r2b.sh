#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r2b
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
echo "== power probe"; timeout 300 python tools/gemm_power_probe.py > $OUT/power_probe.txt 2>&1; cat $OUT/power_probe.txt | grep -v amdgpu.ids
echo "== A/B"; timeout 400 python tools/gemm_ab.py 7 > $OUT/gemm_ab.txt 2>&1; grep -v amdgpu.ids $OUT/gemm_ab.txt
echo "== 2-rank bench (gloo, shared device)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --share-device --backend gloo --batch 128 > $OUT/bench_2rank_shared.json 2> $OUT/bench_2rank.err; tail -1 $OUT/bench_2rank_shared.json
