"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/fsmi355.h (the drop-in boundary) and
include/fsmi355_internal.h (measurement hooks, A/B switches, test read-backs) declare."""
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
    public = [n for n in _declared("fsmi355.h", "fs_") if n != "fs_done_cb"]
    internal = _declared("fsmi355_internal.h", "fs_")
    assert len(public) >= 25 and not set(public) & set(internal)
    for n in public + internal:
        assert hasattr(lib, n), n
    assert sorted(public + internal) == sorted(_capi.RENDER_SYMBOLS)


def test_the_boundary_header_carries_no_laboratory():
    """include/fsmi355.h is what a FractalShark maintainer binds: every probe, A/B switch and statistics read-back lives in
    fsmi355_internal.h, and the reference-language host side (csrc/gpu_render_shim.hpp) needs the public header only."""
    public = set(_declared("fsmi355.h", "fs_"))
    for probe in ("fs_test_block_threshold", "fs_seq_cursor_probe", "fs_read_stats_raw", "fs_time_render_current",
                  "fs_read_tile_order", "fs_set_kernel_variant", "fs_enable_step_count", "fs_read_step_count",
                  "fs_kernel_ms_history", "fs_last_kernel_ms", "fs_read_tile_costs", "fs_forget_tile_costs"):
        assert probe not in public, probe
    shim = open(os.path.join(ROOT, "fractalshark_amd", "csrc", "gpu_render_shim.hpp")).read()
    assert "fsmi355_internal.h" not in shim
    used = set(re.findall(r"\b(fs_[a-z_0-9]+)\s*\(", re.sub(r"//.*", "", shim)))
    assert used and used <= public, sorted(used - public)


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
