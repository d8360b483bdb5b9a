"""Build libmlpk.so (the C-ABI kernel library of include/mlpk.h) for gfx950 with hipcc.

In-tree build: objects under jittor-mlp_amd/build/, the shared library at
jittor-mlp_amd/lib/libmlpk.so (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only authoring container.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmlpk.so")
SOURCES = ["mlpk_gemm.hip", "mlpk_gemm_q4.hip", "mlpk_gemm_skinny.hip", "mlpk_norm.hip", "mlpk_embed.hip", "mlpk_remap.hip", "mlpk_tokenmlp.hip", "mlpk_tokenmlp_t4.hip", "mlpk_dwconv.hip", "mlpk_hire.hip", "mlpk_asconv.hip", "mlpk_chanmlp.hip", "mlpk_backward.hip", "mlpk_vipbranch.hip", "mlpk_smlp.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
FLAGS += os.environ.get("MLPK_EXTRA_FLAGS", "").split()      # tuning aid: A/B builds of a kernel variant (-DTM_...)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


GEN = os.path.join(CSRC, "gen")
GEN_OUT = os.path.join(CSRC, "gen_out")


def _generate():
    """run the kernel generators (csrc/gen/*.py -> csrc/gen_out/*.inc, git-ignored: the generators are the source)"""
    os.makedirs(GEN_OUT, exist_ok=True)
    for gen, inc in (("q4gen.py", "q4_kernels.inc"), ("t4gen.py", "t4_kernels.inc")):
        out = os.path.join(GEN_OUT, inc)
        srcs = [os.path.join(GEN, f) for f in (gen, "q4gen.py", "isa.py")]
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(x) for x in srcs):
            continue
        r = subprocess.run([sys.executable, os.path.join(GEN, gen), out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n" % gen + r.stderr[-4000:])
        os.utime(out)       # emit() leaves an unchanged .inc untouched: mark it current, or every later build() re-runs the generator


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    d += [os.path.join(GEN, f) for f in os.listdir(GEN) if f.endswith(".py")]
    d.append(os.path.join(os.path.dirname(HERE), "include", "mlpk.h"))
    return d


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libmlpk.so.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "stamp.txt")      # next to the library: build/ (386 MB of -save-temps output) does not travel to the GPU box
    digest = _digest(_deps())                      # (sources and generators: not the generated .inc files, which do not travel either)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    _generate()
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        # -save-temps=obj keeps the gfx950 assembly next to the object: tools/isa_lint.py (tests/test_host_cpu.py) checks
        # the hand-scheduled loops in what hipcc actually generated
        cmd = [hipcc] + FLAGS + ["-save-temps=obj", "-Wno-inline-asm", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
