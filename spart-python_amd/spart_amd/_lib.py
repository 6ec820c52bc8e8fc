"""ctypes binding of libspart_hip.so (include/spart_hip.h).  There is no CPU path: if the
library or a GPU is missing every compute entry point raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(HERE, "..", "libspart_hip.so"))

SPART_F32, SPART_F64 = 0, 1
NPARAM, NCOEF, NWL, NWLS, NLINCL = 27, 48, 2001, 2162, 13
NLAYERS = 60            # SPART_NLAYERS: CanopyStructure's default (sailh.py:345)
ABI_VERSION = 6         # SPART_ABI_VERSION of include/spart_hip.h this binding was written against

c_dp = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class SpartTables(ctypes.Structure):
    _fields_ = [(n, c_dp) for n in ("nr", "Kab", "Kca", "Kdm", "Kw", "Ks", "Kant", "cbc", "prot", "GSV", "nw", "Ea")] + [
        ("nb", ctypes.c_int32), ("wl_smac", c_dp), ("coef", c_dp), ("nsrf", ctypes.c_int32), ("wl_srf", c_dp),
        ("p_srf", c_dp)]


class SpartMaterialize(ctypes.Structure):
    _fields_ = [(n, vp) for n in ("leaf_refl", "leaf_tran", "leaf_kchl", "soil_refl", "soil_refl_dry", "rso", "rdo",
                                  "rsd", "rdd", "rsoil", "La", "rdry_in", "band_mean")] + [("prune_unused_bands", ctypes.c_int32),
                                                                                          ("f32_columns", ctypes.c_int32),
                                                                                          ("f32_bands", ctypes.c_int32),
                                                                                          ("fast_prelude", ctypes.c_int32),
                                                                                          ("lidf_in", vp),
                                                                                          ("nlayers", ctypes.c_int32)]


# name -> (restype, argtypes): every symbol include/spart_hip.h declares
SIGNATURES = {
    "spart_ctx_create": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(SpartTables)]),
    "spart_ctx_destroy": (ctypes.c_int, [vp]),
    "spart_last_error": (ctypes.c_char_p, [vp]),
    "spart_build_id": (ctypes.c_char_p, []),
    "spart_abi_version": (ctypes.c_int, []),
    "spart_ctx_nb": (ctypes.c_int, [vp]),
    "spart_ctx_econv": (ctypes.c_int, [vp, c_dp]),
    "spart_ctx_set_row_pitch": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.c_int64]),
    "spart_calculate_tav": (ctypes.c_int, [ctypes.c_double, c_dp, ctypes.c_int64, c_dp]),
    "spart_workspace_bytes": (ctypes.c_size_t, [vp, ctypes.c_int, ctypes.c_int64]),
    "spart_prospect_batch": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(vp), vp, vp, vp, vp,
                                            ctypes.c_size_t, vp]),
    "spart_bsm_batch": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(vp), vp, vp, vp, vp,
                                       ctypes.c_size_t, vp]),
    "spart_lidf_batch": (ctypes.c_int, [vp, ctypes.c_int64, vp, vp, vp, vp]),
    "spart_sailh_batch": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, vp, vp, vp, ctypes.POINTER(vp),
                                         ctypes.POINTER(vp), vp, ctypes.c_int32, ctypes.POINTER(vp), vp, ctypes.c_size_t, vp]),
    "spart_smac_batch": (ctypes.c_int, [vp, ctypes.c_int64, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        vp, ctypes.c_size_t, vp]),
    "spart_run_batch": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(vp), vp, vp, vp, vp, vp,
                                       ctypes.POINTER(SpartMaterialize), vp, ctypes.c_size_t, vp]),
    "spart_lut_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]),
    "spart_lut_nearest": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, vp, ctypes.c_int64, vp, vp, vp, vp,
                                         vp, ctypes.c_size_t, vp]),
    "spart_lut_stats": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, vp,
                                       ctypes.POINTER(ctypes.c_int64), c_dp]),
    "spart_profile_enable": (ctypes.c_int, [vp, ctypes.c_int]),
    "spart_profile_read": (ctypes.c_int, [vp, c_dp, ctypes.POINTER(ctypes.c_int)]),
    "spart_profile_read_stages": (ctypes.c_int, [vp, c_dp, ctypes.POINTER(ctypes.c_int)]),
}
STAGES = ("prelude", "bands", "columns")        # spart_profile_read_stages

_libs = {}


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_spart_build", os.path.join(HERE, "..", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def expected_build_id():
    """id of the sources next to this package (build.py: source_id) -- what the in-tree library must report"""
    return _build_module().source_id(os.environ.get("SPART_FAST_MATH", "1") == "1")


def build_id(lib=None):
    """spart_build_id() of the loaded library"""
    return (lib or load()).spart_build_id().decode()


def load(path=None):
    """dlopen the library (torch is imported first so that its HIP runtime, soname
    libamdhip64.so.7, is the one both sides use).  ``path`` selects another build of the same
    ABI (tools/ab_bench.py compares kernel variants in one process).  The in-tree library must have been built from
    the sources next to it (content hash embedded at build time): a stale one is rebuilt when hipcc is there, and
    refused otherwise -- never loaded silently."""
    path = os.path.abspath(path or os.environ.get("SPART_HIP_LIB") or LIB_PATH)
    if path in _libs:
        return _libs[path]
    import torch  # noqa: F401

    in_tree = path == os.path.abspath(LIB_PATH)
    want = None
    if in_tree:
        try:
            b = _build_module()
            want = expected_build_id()
        except OSError:
            # a deployment that ships the prebuilt library without csrc/: nothing to compare the binary with
            b, want = None, None
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing and the kernel sources to build it from are not installed. "
                                   "spart_amd has no CPU fallback.") from None
        have = b.binary_id(path) if b is not None else None
        if want is not None and have != want:
            # the .so is a build artefact (git-ignored): compile it when it is missing or stale and hipcc is around.  Several
            # ranks may get here at once: one compiles (file lock), the others wait and find the library current.
            try:
                import fcntl
                with open(path + ".lock", "w") as lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    try:
                        if b.binary_id(path) != want:
                            b.build(verbose=True)
                    finally:
                        fcntl.flock(lock, fcntl.LOCK_UN)
            except Exception as e:      # noqa: BLE001
                if os.path.exists(path):
                    raise RuntimeError(f"{path} was built from other sources (build id {have}, sources {want}) and could "
                                       f"not be rebuilt ({e}); run `python spart-python_amd/build.py`") from e
                raise RuntimeError(f"{path} is missing and could not be built ({e}); run `python spart-python_amd/build.py` "
                                   "(hipcc, gfx950). spart_amd has no CPU fallback.") from e
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python spart-python_amd/build.py` "
            "(hipcc, gfx950). spart_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    # the interface version first: a library built from another round's header (SPART_HIP_LIB=, lib_path=) would otherwise be
    # called with shifted arguments / a shorter spart_materialize
    try:
        lib.spart_abi_version.restype = ctypes.c_int
        have_abi = int(lib.spart_abi_version())
    except AttributeError:
        have_abi = None
    if have_abi != ABI_VERSION:
        raise RuntimeError(f"{path} implements interface version {have_abi if have_abi is not None else '< 6 (no spart_abi_version)'}"
                           f", this binding needs {ABI_VERSION} (include/spart_hip.h: SPART_ABI_VERSION); rebuild it with "
                           "`python spart-python_amd/build.py`")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if in_tree and want is not None and lib.spart_build_id().decode() != want:      # (what the loaded code says, not what the file's bytes said)
        raise RuntimeError(f"{path} reports build id {lib.spart_build_id().decode()}, the sources next to it are {want}")
    _libs[path] = lib
    return lib


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.spart_last_error(ctx)
        raise RuntimeError(f"libspart_hip error {rc}: {msg.decode() if msg else '?'}")
