"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/fsmi355.h declares."""
import ctypes as C
import os
import re

from fractalshark_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_render_lib_exports_every_declared_symbol(native_libs):
    lib = C.CDLL(native_libs.LIB_RENDER)
    names = [n for n in _declared("fsmi355.h", "fs_") if n != "fs_done_cb"]
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(names) == sorted(_capi.RENDER_SYMBOLS)


def test_inputs_lib_exports_every_declared_symbol(native_libs):
    lib = C.CDLL(native_libs.LIB_INPUTS)
    for n in _declared("fs_inputs.h", "fsh_"):
        assert hasattr(lib, n), n


def test_error_strings(native_libs):
    lib = _capi.render_lib()
    assert b"antialiasing" in lib.fs_error_string(10002)
    assert lib.fs_error_string(0) == b"no error"


def test_layout_sizes():
    assert C.sizeof(_capi.AtHdr32) == 116
    assert C.sizeof(_capi.Reduction) == 24
    assert C.sizeof(_capi.CplxHdr32) == 12 and C.sizeof(_capi.RealHdr32) == 8
