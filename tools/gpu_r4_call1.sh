#!/bin/bash
# round-4 first GPU call: instruction-slot microbenchmarks, LDS-DMA offset probe, vendor yardstick, Infinity-Cache probe, baseline bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r4c1; mkdir -p $OUT
timeout 60 ./tools/ubench/lds_dma_offset.bin > $OUT/lds_dma_offset.txt 2>&1; cat $OUT/lds_dma_offset.txt
timeout 300 ./tools/ubench/q4_slots_r4.bin > $OUT/q4_slots_r4.txt 2>&1; cat $OUT/q4_slots_r4.txt
timeout 300 python tools/blas_ref.py > $OUT/blas_ref.txt 2>&1; cat $OUT/blas_ref.txt
timeout 300 python tools/mall_probe.py > $OUT/mall_probe.txt 2>&1; cat $OUT/mall_probe.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
