#!/bin/bash
# PMC passes over one model bench: instruction mix, waits, LDS conflicts, MFMA busy per kernel (separate rocprofv3 --pmc runs, kernel-trace only).
# usage: bash tools/pmc_model.sh <model> [kernel-substring]  -> gpurun_out/pmc_<model>/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
m=${1:-convmixer_1536_20}; pat=${2:-}
OUT=$PWD/gpurun_out/pmc_$m
rm -rf $OUT; mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o run -- python $OLDPWD/bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/$tag.json 2> $OUT/$tag.err )
  echo "$tag rc=$?"
done
python - "$OUT" "$pat" <<'PY'
import csv, glob, collections, sys
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if pat and pat not in n:
            continue
        k = (n[:60], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for (n, c), (cnt, s) in sorted(agg.items()):
        if cnt >= 3:
            fo.write("%-60s %-28s launches=%d mean=%.5g\n" % (n, c, cnt, s / cnt))
print(open(out + "/summary.txt").read())
PY
