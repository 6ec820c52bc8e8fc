"""Build libspart_hip.so (the C-ABI library of include/spart_hip.h) in-tree with hipcc for gfx950.

    python spart-python_amd/build.py [--force]      (SPART_FAST_MATH=0 builds the IEEE-division / libm variant)

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with
the source tree (it is git-ignored, not gpurun-ignored).
"""
import hashlib
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


BUILD_ID_TAG = b"SPART_BUILD_ID:"       # csrc/spart_capi.hip embeds TAG + id; spart_build_id() returns the id


def source_id(fast_math=True, extra=()):
    """12 hex digits over everything that determines the binary: the kernel / ABI sources, the compiler flags and the
    math variant.  hipcc gets it as -DSPART_BUILD_ID and spart_build_id() returns it, so a loaded .so can be tied to
    the sources next to it (spart_amd._lib.load refuses a stale one; bench.py prints it)."""
    h = hashlib.sha256()
    for d in sorted(DEPS, key=os.path.basename):
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    h.update(" ".join(FLAGS + list(extra)).encode() + (b"|fast" if fast_math else b"|ieee"))
    return h.hexdigest()[:12]


def binary_id(path=None):
    """The id embedded in a built library, read from the file's bytes (no dlopen, no GPU); None if there is none."""
    try:
        data = open(path or OUT, "rb").read()
    except OSError:
        return None
    i = data.find(BUILD_ID_TAG)
    if i < 0:
        return None
    j = i + len(BUILD_ID_TAG)
    return data[j:j + 12].decode("ascii", "replace")


def needs_build(fast_math=True):
    """True unless the library next to this file was built from exactly the current sources / flags (content hash, not
    modification times: a checkout or a copy to the GPU box resets those)."""
    return binary_id() != source_id(fast_math)


def build(force=False, fast_math=None, verbose=True, out=None, extra=()):
    """out / extra: build a variant (other output path, extra hipcc flags) for tools/ab_bench.py."""
    if out is not None:
        cmd = [hipcc(), *FLAGS, "-o", out, SRC, "-DSPART_FAST_MATH=1", f'-DSPART_BUILD_ID="{source_id(True, extra)}"', *extra]
        if verbose:
            print("[spart_amd] " + " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd)
        return out
    if fast_math is None:
        fast_math = os.environ.get("SPART_FAST_MATH", "1") == "1"   # default: hardware rcp/exp/log/sqrt (parity-tested)
    if not force and not needs_build(fast_math):
        return OUT
    tmp = f"{OUT}.tmp.{os.getpid()}"      # several ranks may get here at once: each links its own file, the rename is atomic
    cmd = [hipcc(), *FLAGS, f'-DSPART_BUILD_ID="{source_id(fast_math)}"', "-o", tmp, SRC]
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
