#!/bin/bash
# final measurement run of the round: everything the judge reads, from one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
python __graft_entry__.py build > $OUT/build.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/rocminfo.txt; nproc >> $OUT/rocminfo.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
echo "== bs=256 parity (printed distances)"; timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -s -k "batch_256" > $OUT/bs256_parity.txt 2>&1; tail -3 $OUT/bs256_parity.txt
echo "== pmc"; bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; tail -2 $OUT/pmc.log; cp gpurun_out/pmc/summary.txt $OUT/pmc_summary.txt; cp gpurun_out/pmc/traffic.json $OUT/traffic.json
mkdir -p profiles; cp $OUT/traffic.json profiles/${ROUND:-r05}_traffic.json
echo "== bench (default)"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d['cpu_baseline']['config1']['value'])"
echo "== 2 ranks on one device"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-device --steps 20 --warmup 5 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; tail -c 600 $OUT/bench_2rank.json
timeout 300 bash tools/prof_model.sh mixer_b16 < /dev/null 2>&1 | tail -12
: > $OUT/bench_models.jsonl
for m in mixer_s16 mixer_l16 gmlp_s resmlp_24 vip_s7 s2mlpv2 asmlp_t convmixer_1536_20 sparsemlp_t hiremlp_s msmlp_t swinmlp_t cyclemlp_b1; do timeout 200 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline < /dev/null >> $OUT/bench_models.jsonl 2>> $OUT/bench_models.err; done
python - <<'PY'
import json
for l in open("gpurun_out/final/bench_models.jsonl"):
    d = json.loads(l)
    print("%-40s %10.1f img/s %8.2f ms  %7.1f model-TF/s" % (d["metric"], d["value"], d["ms_per_step"], d["model_tflops"]))
PY
for m in vip_s7 gmlp_s resmlp_24 asmlp_t convmixer_1536_20 msmlp_t swinmlp_t hiremlp_s cyclemlp_b1; do timeout 200 bash tools/prof_model.sh $m < /dev/null 2>&1 | tail -9; done
if [ -n "$FINAL_FULL" ]; then
echo "== torch / runtime kernels per forward (one-time packing separated)"; timeout 400 bash tools/prof_aten.sh mixer_b16 < /dev/null 2>&1 | tail -12
echo "== epilogue statistics A/B"; timeout 400 bash tools/gpu_stats.sh < /dev/null > $OUT/epilogue_stats_ab.txt 2>&1; cat $OUT/epilogue_stats_ab.txt
fi
