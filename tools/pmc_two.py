"""The program tools/pmc_ab.sh points rocprofv3 at: full float32 steps of several library builds in a FIXED alternating
order (a, b, ..., a, b, ...; no shuffling, so that dispatch k of a kernel belongs to build k mod n).

    python3 tools/pmc_two.py a=build_ab/a.so b=build_ab/b.so [--reps 3]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))


def main():
    builds = [a for a in sys.argv[1:] if "=" in a]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
    import torch
    from spart_amd import workloads
    from spart_amd.engine import Engine
    P = torch.as_tensor(workloads.lhs_params(1_000_000, "full").T.copy(), device="cuda:0")
    os.environ["SPART_SIDE_STREAM"] = "0"          # every kernel on one stream: the counters of k_bands are its own
    engs = [Engine("Sentinel2A-MSI", 0, lib_path=b.split("=", 1)[1]) for b in builds]
    for _ in range(reps):
        for e in engs:
            e.run(P, "float32")
            torch.cuda.synchronize()


if __name__ == "__main__":
    main()
