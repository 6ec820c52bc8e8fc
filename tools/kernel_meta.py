"""Register / scratch / LDS budget of every kernel in libspart_hip (hipcc -S of the device code; no GPU needed).

    python tools/kernel_meta.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
import build  # noqa: E402


def kernel_meta(extra=()):
    with tempfile.TemporaryDirectory() as d:
        meta, cur = {}, None
        for line in (l for f in build.device_asm(d, extra) for l in open(f)):     # every translation unit, with ITS flags
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                cur = m.group(1)
                meta[cur] = {}
                continue
            m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|"
                         r"group_segment_fixed_size):\s+(\d+)", line)
            if m and cur:
                meta[cur][m.group(1)] = int(m.group(2))
    return {k: v for k, v in meta.items() if "vgpr_count" in v}


if __name__ == "__main__":
    names = kernel_meta(sys.argv[1:])
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for d, (k, v) in zip(dem, names.items()):
        d = re.sub(r"\(.*", "", d).replace("void spart::", "")
        print(f"{d:60s} vgpr {v['vgpr_count']:4d} sgpr {v['sgpr_count']:4d} scratch {v['private_segment_fixed_size']:5d} "
              f"lds {v.get('group_segment_fixed_size', 0):6d} spill v{v['vgpr_spill_count']} s{v['sgpr_spill_count']}")
