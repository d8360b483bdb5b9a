#!/bin/bash
# One PMC pass (LDS bank conflicts) over several model benches: finds kernels whose LDS accesses serialise on a few banks (round 6: the Swin-MLP
# window kernel's loader sat on two banks per wave).  usage: bash tools/pmc_lds_conflicts.sh model [model ...] -> gpurun_out/pmc_lds/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
for m in "$@"; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$m -o run -- python $OLDPWD/bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/$m.json 2> $OUT/$m.err )
  echo "$m rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fo:
    for d in sorted(glob.glob(out + "/*/")):
        model = os.path.basename(d.rstrip("/"))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT": cnt[r["Kernel_Name"][:70]] += 1
        fo.write("== %s\n" % model)
        rows = []
        for k, c in agg.items():
            act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
            if act <= 0: continue
            rows.append((c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, k, cnt[k], c.get("SQ_LDS_BANK_CONFLICT", 0.0), act, c.get("SQ_BUSY_CYCLES", 0.0)))
        for ratio, k, n, bc, act, busy in sorted(rows, reverse=True)[:12]:
            fo.write("%-70s launches=%4d  conflict/active=%.2f  lds_active/busy=%.3f\n" % (k, n, ratio, act / busy if busy else 0))
print(open(out + "/summary.txt").read())
PY
