"""Build libcosyvoice_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcosyvoice_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-value",
         "-Wno-pass-failed"]
# per-file additions.  attention.hip: no NaN is ever formed on its softmax path (masked scores are a large finite negative),
# and without the flag every v_max on an MFMA result is preceded by a canonicalizing v_max(x, x).
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"]}


def _needs(obj, srcs):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src):
    obj = src[:-4] + ".o"
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if _needs(obj, deps):
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-8000:]))
        return obj, True
    return obj, False


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if force:
        for s in srcs:
            o = s[:-4] + ".o"
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or _needs(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("built", LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
