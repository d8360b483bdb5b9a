#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) + duration of every kernel of several models: which memory-bound
# kernels move more bytes than they must, or move them slowly.  usage: bash tools/pmc_traffic_models.sh model ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for m in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$m/$c -o run -- python $OLDPWD/bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/$m.$c.json 2> $OUT/$m.$c.err )
  done
  python - "$OUT/$m" "$m" <<'PY' | tee $OUT/$m.txt
import csv, glob, collections, sys
out, m = sys.argv[1], sys.argv[2]
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
    if n not in dur or dur[n][0] < 2: continue
    us = dur[n][1] / dur[n][0]
    f = c["FETCH_SIZE"][1] / max(1, c["FETCH_SIZE"][0]); w = c["WRITE_SIZE"][1] / max(1, c["WRITE_SIZE"][0])
    rows.append((dur[n][1], dur[n][0], us, f / 1e3, w / 1e3, (f + w) * 1024 / us / 1e6, n[:90]))
print("== %s   (total us, calls, avg us, fetch MB, write MB, (fetch+write)/time TB/s [raw counter units: KB], kernel)" % m)
for r in sorted(rows, reverse=True)[:14]:
    print("%9.0f %5d %8.1f %9.1f %9.1f %6.2f  %s" % r)
PY
done
