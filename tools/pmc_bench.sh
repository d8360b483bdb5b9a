#!/bin/bash
# PMC passes over the default bench (separate rocprofv3 --pmc runs, kernel-trace only: MI355X_MICROARCH.md HBM recipe).
# usage (on the GPU box): bash tools/pmc_bench.sh   -> gpurun_out/pmc/summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants --streams 1 > $OUT/$tag.json 2> $OUT/$tag.err )
  echo "$tag rc=$?"
done
python - <<'PY'
import csv, glob, collections, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/pmc")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"mlpk(\d+)(\w+?)I", name)
        short = re.sub(r"^_ZN4mlpk\d+", "", name).split("I")[0] if name.startswith("_ZN4mlpk") else name[:40]
        if "gemm_nt" in name:
            short += "<bf16>" if "DF16b" in name else ""
        k = (short, r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc <group> --kernel-trace (separate passes) on: python bench.py --steps 3 --warmup 1 (Mixer-B/16, bs=256, bf16)\n")
    fo.write("# per-launch means; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: wide coalesced reads are reported at HALF their bytes -> x2, MI355X_MICROARCH.md)\n")
    for (short, c), (n, s) in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        if not any(t in short for t in ("gemm_nt", "q4_", "t4_", "token_mlp", "row_stats", "norm_apply", "layernorm_transpose", "stats_finalize")):
            continue
        fo.write("%-40s %-30s launches=%d mean=%.5g\n" % (short, c, n, s / n))
print(open(out + "/summary.txt").read())
# traffic of the dominant kernel for bench.py's roofline.traffic, stamped with the source it was measured on.
# One mlpk_gemm_nt call of the persistent tile is up to three launches (one per tile height, p8_plan): the figure is per CALL =
# all bytes of all gemm_nt_p8 launches / the calls that made them (4 forwards x (24 channel-MLP GEMMs + the patch embedding)).
import hashlib, json, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
full = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("gemm_nt_p8" in r["Kernel_Name"] or r["Kernel_Name"].startswith("q4_")) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = (r["Kernel_Name"], r["Counter_Name"])
            full[k][0] += 1
            full[k][1] += float(r["Counter_Value"])
if full:
    calls = 4 * 25
    fetch = sum(v[1] for k, v in full.items() if k[1] == "FETCH_SIZE")
    write = sum(v[1] for k, v in full.items() if k[1] == "WRITE_SIZE")
    head = open(os.path.join(root, "tools", ".git_head")).read().strip() if os.path.exists(os.path.join(root, "tools", ".git_head")) else "?"
    tj = {"channel_mlp_gemm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / calls,
          "per_instantiation_kib_per_launch": {k[0][-40:] + " " + k[1]: [v[0], v[1] / v[0]] for k, v in sorted(full.items())},
          "gemm_calls": calls, "git": head,
          "gemm_source_sha256": __import__("bench").gemm_source_digest(),
          "note": "bytes per mlpk_gemm_nt CALL of the channel-MLP GEMMs in bench.py --steps 3 --warmup 1: all launches of the generated q4 kernel (fc1) and of gemm_nt_p8_kernel (fc2 and the patch embedding; one launch per tile height) summed, (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md), divided by 4 forwards x 25 calls (24 channel-MLP GEMMs + the patch embedding, which has fc2's shape)"}
    json.dump(tj, open(out + "/traffic.json", "w"))
    print(json.dumps(tj))
PY
