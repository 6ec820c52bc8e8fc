"""Import the upstream reference package (read-only tree at /root/reference) in THIS container.

Only used by the table exporter and the golden-vector generator; nothing that runs on the
GPU box imports this.  The reference hard-imports ``nvtx`` (SPART.py:23) which is not
installed here, so a no-op stand-in for that *profiling hook only* is registered first
(SURVEY.md §8c).  The reference tree itself is never modified or copied.
"""
import contextlib
import sys
import types
import warnings

REFERENCE_SRC = "/root/reference/src"


def import_reference():
    if "nvtx" not in sys.modules:
        nv = types.ModuleType("nvtx")

        class annotate(contextlib.ContextDecorator):
            def __init__(self, *a, **k):
                pass

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        nv.annotate = annotate
        sys.modules["nvtx"] = nv
    # make sure no alias package named SPART from this repo shadows the reference
    for p in list(sys.path):
        if p.rstrip("/").endswith("spart-python_amd"):
            sys.path.remove(p)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    warnings.filterwarnings("ignore")
    import SPART  # noqa

    assert SPART.__file__.startswith(REFERENCE_SRC), SPART.__file__
    return SPART
