#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) + duration per kernel of ONE command: bash tools/pmc_one.sh LABEL 'command'  -> gpurun_out/pmc_one/LABEL.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
L=$1; shift
OUT=$PWD/gpurun_out/pmc_one/$L; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o run -- bash -c "cd $OLDPWD && $*" > $OUT/$c.out 2> $OUT/$c.err )
done
python - "$OUT" <<'PY' | tee $OUT.txt
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for f in glob.glob(out + "/FETCH_SIZE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur[r["Kernel_Name"]]
        d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for n, c in agg.items():
    if n not in dur: continue
    us = dur[n][1] / dur[n][0]
    f = c["FETCH_SIZE"][1] / max(1, c["FETCH_SIZE"][0]); w = c["WRITE_SIZE"][1] / max(1, c["WRITE_SIZE"][0])
    rows.append((dur[n][1], dur[n][0], us, f / 1e3, w / 1e3, n[:100]))
print("total us, calls, avg us, fetch MB, write MB (raw counter KB / 1e3, averaged over calls), kernel")
for r in sorted(rows, reverse=True)[:16]:
    print("%9.0f %5d %8.1f %9.1f %9.1f  %s" % r)
PY
find $OUT -name "*.csv" -delete
