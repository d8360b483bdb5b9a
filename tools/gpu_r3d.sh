#!/bin/bash
# kernel-trace of the default bench: where does the step time go between kernels?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py build > /dev/null 2>&1
OUT=$PWD/gpurun_out/r3d; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/bench.json 2> $OUT/err.txt )
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3d/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last 3 forwards: find the last ~ (3 * kernels per forward) kernels; a forward starts with the patchify kernel
starts = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
a, b = starts[-3], len(rows)
seg = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
print("kernels in 3 forwards: %d, span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%)" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
import statistics
pos = [g for g in gaps if g > 0]
print("gaps: n %d median %.2f us mean %.2f us max %.1f us; overlaps (negative gaps): %d" % (len(pos), statistics.median(pos) / 1e3, sum(pos) / len(pos) / 1e3, max(pos) / 1e3, sum(1 for g in gaps if g <= 0)))
big = sorted(((g, seg[i]["Kernel_Name"][:50], seg[i + 1]["Kernel_Name"][:50]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, k0, k1 in big: print("  %.1f us between %s -> %s" % (g / 1e3, k0, k1))
PY
cat $OUT/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
find $OUT -name "*kernel_trace.csv" -delete
