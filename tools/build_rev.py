"""Build libspart_hip.so from a git revision (for A/B runs against the working tree): build_rev.py <rev> <out.so>
Files added after <rev> (ABI growth) are taken from the working tree, only csrc/spart_math.h and spart_kernels.h may
come from the old revision: --files a,b selects which."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rev, out = sys.argv[1], os.path.abspath(sys.argv[2])
files = sys.argv[3].split(",") if len(sys.argv) > 3 else ["spart-python_amd/csrc/spart_math.h"]
tmp = tempfile.mkdtemp()
shutil.copytree(os.path.join(ROOT, "spart-python_amd", "csrc"), os.path.join(tmp, "spart-python_amd", "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
for f in files:
    data = subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{f}"])
    open(os.path.join(tmp, f), "wb").write(data)
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import build
build.SOURCES = [os.path.join(tmp, "spart-python_amd", "csrc", os.path.basename(s)) for s in build.SOURCES]   # both translation units
build._compile_and_link(out, ["-DSPART_FAST_MATH=1", f'-DSPART_BUILD_ID="rev-{rev[:8]}"'], (), True)
shutil.rmtree(tmp)
print(out)
