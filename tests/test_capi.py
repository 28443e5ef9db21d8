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


def test_loading_the_library_asks_for_dmabuf_ipc_without_overriding_the_host():
    """RCCL between the members of an fs_group (and between bench.py's ranks) needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool's
    hosts, and the ROCm runtime reads it once when it comes up: the library sets it when it is LOADED (a constructor in
    csrc/group.cpp), unless the host application chose a value itself.  Checked in fresh processes."""
    import subprocess
    import sys
    prog = ("import os, ctypes, sys; sys.path.insert(0, %r); from fractalshark_amd import _build; ctypes.CDLL(_build.LIB_RENDER); "
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p; print(g(b'HSA_ENABLE_IPC_MODE_LEGACY').decode())" % ROOT)
    for preset, want in ((None, "0"), ("1", "1")):
        env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
        if preset is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = preset
        out = subprocess.run([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, check=True).stdout.decode().strip()
        assert out == want, (preset, out)


def test_bench_rank_sets_dmabuf_ipc_before_the_runtime_comes_up():
    """A rank started by somebody else's torchrun never passes through bench.launch_ranks: main() itself must set the variable,
    and before torch is imported."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    at = main.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")')
    assert at < main.index("import torch")
