"""The C-ABI library must load (no GPU needed) and export every symbol include/clstm_abi.h
declares; the ctypes table in clstm_amd/abi.py must cover the same set.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "clstm_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clstm_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def libpath():
    path = os.path.join(ROOT, "clstm_amd", "lib", "libclstm_hip.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "clstm_amd", "csrc"), "-s", "all"])
    return path


def test_header_and_binding_agree():
    from clstm_amd import abi
    assert header_symbols() == abi.EXPORTED_SYMBOLS


def test_library_exports_every_declared_symbol(libpath):
    dll = ctypes.CDLL(libpath)
    for name in header_symbols():
        assert hasattr(dll, name), name
    dll.clstm_abi_version.restype = ctypes.c_int
    assert dll.clstm_abi_version() == 1


def test_library_is_gfx950_code_object(libpath):
    """The fat binary embedded in the shared object must carry a gfx950 code object (and only that GPU target)."""
    blob = open(libpath, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_missing_library_fails_loudly(tmp_path):
    from clstm_amd import abi
    with pytest.raises(abi.ClstmError):
        abi.Lib(str(tmp_path / "libclstm_hip.so"))
