#!/bin/bash
# rocprofv3 kernel stats of one model bench: bash tools/prof_model.sh <model> -> gpurun_out/prof_<model>/..._kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
m=$1
OUT=$PWD/gpurun_out/prof_$m
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $m -- python $OLDPWD/bench.py --model $m --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/bench.json 2> $OUT/err.txt )
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over 4 forwards -> %.2f ms/forward" % (tot / 1e6, tot / 4e6))
for r in rows[:14]:
    n = r["Name"]
    n = re.sub(r"^_ZN4mlpk\d+", "", n)[:70]
    print("%-70s calls %5s  avg %9.1f us  %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find $OUT -name "*kernel_trace.csv" -delete
