"""CPU-only: the reference-side binding (gpu_render_shim.hpp: GPURenderer members forwarding to the C ABI)
compiles against stand-ins of the reference types and links against libfsmi355.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_compiles_and_links(native_libs, tmp_path):
    exe = tmp_path / "shim_selftest"
    cmd = ["g++", "-std=c++17", "-Wall", "-Wno-unused-private-field", "-o", str(exe),
           os.path.join(ROOT, "tests", "shim_selftest.cpp"), native_libs.LIB_RENDER,
           "-Wl,-rpath," + os.path.dirname(native_libs.LIB_RENDER), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
           "-lamdhip64", "-pthread"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
