"""The C-ABI library builds for gfx950 (hipcc cross-compiles without a GPU), loads, and exports
every function include/spart_hip.h declares.  No compute calls here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    sys.path.insert(0, os.path.join(ROOT, "spart-python_amd"))
    import build
    return build.build(verbose=False)


def declared_functions():
    src = open(os.path.join(ROOT, "include", "spart_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spart_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    names = declared_functions()
    for n in ("spart_ctx_create", "spart_ctx_destroy", "spart_last_error", "spart_workspace_bytes",
              "spart_prospect_batch", "spart_bsm_batch", "spart_lidf_batch", "spart_sailh_batch", "spart_smac_batch",
              "spart_run_batch"):
        assert n in names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/spart_hip.h but not exported"


def test_ctypes_signatures_cover_the_header(lib_path):
    from spart_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()


def test_abi_version_matches_header_and_binding(lib_path):
    """spart_abi_version() == SPART_ABI_VERSION of include/spart_hip.h == the ctypes binding's; the loader refuses a library
    that reports another one (a build of an earlier round loaded through SPART_HIP_LIB would be called with shifted arguments)"""
    from spart_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "spart_hip.h")).read()
    v = int(re.search(r"#define\s+SPART_ABI_VERSION\s+(\d+)", hdr).group(1))
    lib = ctypes.CDLL(lib_path)
    assert lib.spart_abi_version() == v == _lib.ABI_VERSION
    # the struct the binding hands over has the header's members, in order
    members = re.search(r"typedef struct spart_materialize \{(.*?)\} spart_materialize;", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S), flags=re.S).group(1)
    names = re.findall(r"\*?\b([A-Za-z_0-9]+)\s*[,;]", members)
    assert names == [f[0] for f in _lib.SpartMaterialize._fields_], names


def test_code_object_targets_gfx950(lib_path):
    data = open(lib_path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data


def test_null_context_errors_do_not_need_a_gpu(lib_path):
    lib = ctypes.CDLL(lib_path)
    lib.spart_last_error.restype = ctypes.c_char_p
    lib.spart_workspace_bytes.restype = ctypes.c_size_t
    assert lib.spart_workspace_bytes(None, 0, ctypes.c_int64(10)) == 0
    rc = lib.spart_ctx_create(None, 0, None)
    assert rc == -1 and b"null" in lib.spart_last_error(None)


def test_build_id_ties_the_binary_to_its_sources(lib_path, tmp_path):
    """spart_build_id() == the hash of the sources / flags next to the library (build.source_id), read both from the loaded
    code and from the file's bytes; a file without (or with another) id is detected without loading it."""
    import build
    lib = ctypes.CDLL(lib_path)
    lib.spart_build_id.restype = ctypes.c_char_p
    want = build.source_id(True)
    assert lib.spart_build_id().decode() == want and build.binary_id(lib_path) == want and len(want) == 12
    assert not build.needs_build(True)
    fake = tmp_path / "lib.so"
    fake.write_bytes(b"\x7fELF....SPART_BUILD_ID:0123456789ab....")
    assert build.binary_id(str(fake)) == "0123456789ab" != want
    fake.write_bytes(b"\x7fELF no id here")
    assert build.binary_id(str(fake)) is None and build.binary_id(str(tmp_path / "missing.so")) is None
    assert build.source_id(False) != want and build.source_id(True, ["-DX=1"]) != want      # flags are part of the id
