"""Import spart_amd/workloads.py WITHOUT importing the spart_amd package (CPU tests must not need the HIP library)."""
import importlib.util
import os

_p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spart-python_amd", "spart_amd", "workloads.py")
_s = importlib.util.spec_from_file_location("_spart_amd_workloads", _p)
_m = importlib.util.module_from_spec(_s)
_s.loader.exec_module(_m)
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
