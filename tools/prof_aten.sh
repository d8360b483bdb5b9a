#!/bin/bash
# Which torch (ATen / runtime copy) kernels run PER FORWARD, as opposed to once (weight packing, zero-filled workspaces)?
# Two rocprofv3 --stats runs of the same bench with 2 and 12 timed steps; the difference of their call counts / 10 is the per-forward share.
# usage: bash tools/prof_aten.sh <model>   -> gpurun_out/prof_aten_<model>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
m=${1:-mixer_b16}
for n in 2 12; do
  OUT=$PWD/gpurun_out/prof_aten_${m}_$n; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $m -- python $OLDPWD/bench.py --model $m --steps $n --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/bench.json 2> $OUT/err.txt )
  find $OUT -name "*kernel_trace.csv" -delete
done
python - $m <<'PY' | tee gpurun_out/prof_aten_$1.txt
import csv, glob, sys
m = sys.argv[1]
def load(n):
    f = glob.glob("gpurun_out/prof_aten_%s_%d/**/*kernel_stats.csv" % (m, n), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(2), load(12)
tot = sum(b[k][1] - a.get(k, (0, 0.0))[1] for k in b) / 10
print("%s: kernel time per forward %.3f ms (difference of a 12-step and a 2-step run / 10)" % (m, tot / 1e6))
rows = []
for k in b:
    dc = (b[k][0] - a.get(k, (0, 0))[0]) / 10.0
    dt = (b[k][1] - a.get(k, (0, 0.0))[1]) / 10.0
    mine = not ("at::native" in k or "rocclr" in k or "rocblas" in k or "hipblas" in k or k.startswith("void at::"))
    rows.append((mine, dt, dc, k, a.get(k, (0, 0))[0] - 3 * dc))
print("-- kernels that are NOT this library's (torch / HIP runtime), per forward:")
at = 0.0
for mine, dt, dc, k, once in sorted(rows, key=lambda r: -r[1]):
    if not mine and (dc > 0.05 or once > 0):
        at += dt
        print("   %-80s per forward: %5.1f calls %8.1f us   once-only calls: %d" % (k[:80], dc, dt / 1e3, round(once)))
print("   total per forward: %.1f us = %.2f %% of the kernel time" % (at / 1e3, 100 * at / tot))
PY
