"""Build libspart_hip.so (the C-ABI library of include/spart_hip.h) in-tree with hipcc for gfx950.

    python spart-python_amd/build.py [--force]      (SPART_FAST_MATH=0 builds the IEEE-division / libm variant)

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with
the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "spart_capi.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("spart_kernels.h", "spart_math.h", "spart_e3_coeffs.h")] + [
    os.path.join(HERE, "..", "include", "spart_hip.h")]
OUT = os.path.join(HERE, "libspart_hip.so")
# -fno-slp-vectorize: hipcc's SLP pass packs independent fp32 ops into v_pk_mul/v_pk_fma, which issue at
# half rate on gfx950 and block FMA contraction; the VALU-bound band kernel is 12 % faster without it
# (profiles/README.md)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize", "-ldl"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, fast_math=None, verbose=True, out=None, extra=()):
    """out / extra: build a variant (other output path, extra hipcc flags) for tools/ab_bench.py."""
    if out is not None:
        cmd = [hipcc(), *FLAGS, "-o", out, SRC, "-DSPART_FAST_MATH=1", *extra]
        if verbose:
            print("[spart_amd] " + " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd)
        return out
    if fast_math is None:
        fast_math = os.environ.get("SPART_FAST_MATH", "1") == "1"   # default: hardware rcp/exp/log/sqrt (parity-tested)
    if not force and not needs_build():
        return OUT
    tmp = f"{OUT}.tmp.{os.getpid()}"      # several ranks may get here at once: each links its own file, the rename is atomic
    cmd = [hipcc(), *FLAGS, "-o", tmp, SRC]
    if fast_math:
        cmd.insert(1, "-DSPART_FAST_MATH=1")
    if verbose:
        print("[spart_amd] " + " ".join(cmd), file=sys.stderr, flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, fast_math=True if "--fast-math" in sys.argv else None)
    print(OUT)
