"""CPU-only, needs /root/reference (skipped elsewhere): the reference-side binding gpu_render_shim.hpp compiled against
the reference's REAL headers (GPU_Render.h, LAReference.h, PerturbationResults.h, BLAS.h and everything they pull in)
with EVERY explicit instantiation FractalSharkGpuLib/GPU_Render.cu lists (tests/shim/gpu_render_hip_real.cpp), i.e. the
translation unit a maintainer puts in GPU_Render.cu's place.  Checks:
  * it compiles (clang++ -std=c++23: the reference needs C++23 and GCC rejects its member aliases);
    the only thing written for the compile is a <format> header for libstdc++ 11, which lacks that STANDARD header --
    no reference header is replaced;
  * the object defines all 204 member instantiations of the reference's list;
  * every undefined symbol is either exported by libfsmi355.so, a C/C++ runtime symbol, or a member of a reference
    class that FractalSharkLib itself defines (GrowableVector<...>::GetData/GetSize) -- nothing else is missing at link.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "FractalSharkLib")) and os.path.exists(CLANG)),
                                reason="needs the reference tree and ROCm clang++")

FORMAT_STUB = """#pragma once
// libstdc++ 11 has no <format>; the reference only uses std::format in ToString() helpers
#include <string>
namespace std {
template <class... A> std::string format(const char *, A &&...) { return {}; }
template <class... A> std::string format(const std::string &, A &&...) { return {}; }
}
"""


def test_shim_compiles_against_real_reference_headers_with_full_instantiation_list(native_libs, tmp_path):
    stub = tmp_path / "stdstub"
    stub.mkdir()
    (stub / "format").write_text(FORMAT_STUB)
    obj = tmp_path / "gpu_render_hip_real.o"
    cmd = [CLANG, "-std=c++23", "-fPIC", "-c", "-Wall", "-Wno-unused", "-Wno-unknown-pragmas", "-I" + str(stub),
           "-I/opt/conda/include", "-I" + REF + "/HpSharkFloatLib", "-I" + REF + "/FractalSharkPlatform/Common",
           "-I" + REF + "/FractalSharkLib", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "fractalshark_amd", "csrc"),
           os.path.join(ROOT, "tests", "shim", "gpu_render_hip_real.cpp"), "-o", str(obj)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "error" not in p.stdout

    nm = subprocess.run(["nm", "-C", str(obj)], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    defined = [l for l in nm if re.search(r" [WT] .*GPURenderer::", l)]

    def count(member):
        return sum(1 for l in defined if re.search(r"GPURenderer::%s<" % member, l))

    # GPU_Render.cu:227-230, 409-429, 583-594, 503-537, 849-991, 1192-1300, 1380-1436, 1610-1692
    assert count("ClearMemory") == 2
    assert count("InitializeMemory") == 2
    assert count("RenderCurrent") == 2
    assert count("InitializePerturb") == 24
    assert count("Render") == 16
    assert count("RenderPerturbLAv2") == 72
    assert count("RenderPerturbBLAScaled") == 4
    assert count("RenderPerturbBLA") == 6

    exported = subprocess.run(["nm", "-D", "--defined-only", native_libs.LIB_RENDER], stdout=subprocess.PIPE, text=True,
                              check=True).stdout
    exported = {l.split()[-1] for l in exported.splitlines() if l.strip()}
    undefined = [l.split(" U ", 1)[1].strip() for l in nm if " U " in l]
    fs_syms = [u for u in undefined if u.startswith("fs_")]
    assert fs_syms, "the shim must reach the C ABI"
    assert all(u in exported for u in fs_syms), [u for u in fs_syms if u not in exported]
    runtime = re.compile(r"^(std::|operator |__cxa|__gxx|_Unwind|__stack_chk|mem(cpy|set|move)|__dso_handle|"
                         r"vtable for __cxxabiv1|typeinfo for|__assert_fail|abort|strlen|scalbn|ldexp|frexp|fabs|sqrt)")
    reference_members = re.compile(r"^GrowableVector<.*>::(GetData|GetSize)\(\) const$")
    rest = [u for u in undefined if not u.startswith("fs_") and not runtime.match(u) and not reference_members.match(u)]
    assert rest == [], rest
