"""Build libspart_hip.so (the C-ABI library of include/spart_hip.h) in-tree with hipcc for gfx950.

    python spart-python_amd/build.py [--force]      (SPART_FAST_MATH=0 builds the IEEE-division / libm variant)

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with
the source tree (it is git-ignored, not gpurun-ignored).

The library is built from TWO translation units: csrc/spart_capi.hip (the C ABI and every kernel but one family) and
csrc/spart_bands_f32.hip (the float32 full-band kernels, 89 % of the headline step), each with its own flags (TU_FLAGS).
"""
import hashlib
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SRC = os.path.join(CSRC, "spart_capi.hip")
SOURCES = [SRC, os.path.join(CSRC, "spart_bands_f32.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("spart_kernels.h", "spart_math.h", "spart_e3_coeffs.h", "spart_f64_tables.h",
                                                  "spart_bands_f32.h", "spart_lut.h")] + [os.path.join(HERE, "..", "include", "spart_hip.h")]
OUT = os.path.join(HERE, "libspart_hip.so")
# -fno-slp-vectorize: hipcc's SLP pass packs independent fp32 ops into v_pk_mul/v_pk_fma, which issue at
# half rate on gfx950 and block FMA contraction; the VALU-bound band kernel is 12 % faster without it
# (profiles/README.md)
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]
FLAGS = CFLAGS + LDFLAGS            # (what identifies a build; tools that compile one file to assembly use CFLAGS)
# Per-translation-unit flags.  The float32 full-band kernel runs at the issue rate of its instruction mix, and which
# SCHEDULE of those instructions the compiler picks moves it by several per cent.  LLVM's
# "iterative-minreg" scheduling strategy measured 1.7-2.6 % faster than the default for k_bands<float, ...> in three
# interleaved, order-shuffled A/B runs (tools/ab_bench.py); applied to the whole library it costs the float64 column
# kernels a wave of occupancy (k_slots 92 -> 100 VGPRs, k_sensor spills), hence the separate unit.
TU_FLAGS = {"spart_bands_f32.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


BUILD_ID_TAG = b"SPART_BUILD_ID:"       # csrc/spart_capi.hip embeds TAG + id; spart_build_id() returns the id


def tu_flags(src, extra=()):
    """compile flags of one translation unit (without the math / build-id defines)"""
    return CFLAGS + TU_FLAGS.get(os.path.basename(src), []) + list(extra)


def source_id(fast_math=True, extra=()):
    """12 hex digits over everything that determines the binary: the kernel / ABI sources, the compiler flags (per
    translation unit) and the math variant.  hipcc gets it as -DSPART_BUILD_ID and spart_build_id() returns it, so a loaded
    .so can be tied to the sources next to it (spart_amd._lib.load refuses a stale one; bench.py prints it)."""
    h = hashlib.sha256()
    for d in sorted(DEPS, key=os.path.basename):
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    for s in SOURCES:
        h.update((os.path.basename(s) + ":" + " ".join(tu_flags(s, extra))).encode())
    h.update(" ".join(LDFLAGS).encode() + (b"|fast" if fast_math else b"|ieee"))
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


def device_asm(outdir, extra=()):
    """hipcc -S --cuda-device-only of every translation unit with ITS flags -> list of .s files (tools/kernel_meta.py,
    tools/isa_hist.py, tests/test_kernel_resources.py: register budgets and instruction mixes, no GPU needed)."""
    outs, procs = [], []
    for s in SOURCES:
        o = os.path.join(outdir, os.path.basename(s) + ".s")
        procs.append(subprocess.Popen([hipcc(), *tu_flags(s, extra), "-DSPART_FAST_MATH=1", "-S", "--cuda-device-only", "-o", o, s],
                                      stderr=subprocess.DEVNULL))
        outs.append(o)
    for p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, p.args)
    return outs


def _compile_and_link(out, defines, extra, verbose):
    with tempfile.TemporaryDirectory(prefix="spart_build_") as tmp:
        objs, procs = [], []
        for s in SOURCES:                                   # the units compile side by side
            o = os.path.join(tmp, os.path.basename(s) + ".o")
            cmd = [hipcc(), *tu_flags(s, extra), *defines, "-c", "-o", o, s]
            if verbose:
                print("[spart_amd] " + " ".join(cmd), file=sys.stderr, flush=True)
            procs.append((subprocess.Popen(cmd), cmd))
            objs.append(o)
        for p, cmd in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        cmd = [hipcc(), *LDFLAGS, "-o", out, *objs]
        if verbose:
            print("[spart_amd] " + " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd)


def build(force=False, fast_math=None, verbose=True, out=None, extra=()):
    """out / extra: build a variant (other output path, extra hipcc flags for every unit) for tools/ab_bench.py."""
    if out is not None:
        _compile_and_link(out, ["-DSPART_FAST_MATH=1", f'-DSPART_BUILD_ID="{source_id(True, extra)}"'], extra, verbose)
        return out
    if fast_math is None:
        fast_math = os.environ.get("SPART_FAST_MATH", "1") == "1"   # default: hardware rcp/exp/log/sqrt (parity-tested)
    if not force and not needs_build(fast_math):
        return OUT
    tmp = f"{OUT}.tmp.{os.getpid()}"      # several ranks may get here at once: each links its own file, the rename is atomic
    defines = [f'-DSPART_BUILD_ID="{source_id(fast_math)}"'] + (["-DSPART_FAST_MATH=1"] if fast_math else [])
    try:
        _compile_and_link(tmp, defines, (), verbose)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, fast_math=True if "--fast-math" in sys.argv else None)
    print(OUT)
