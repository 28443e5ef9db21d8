"""Build recipes for the native libraries (in-tree, so the .so files travel to the GPU box).

  csrc/libfsmi355.so   hipcc --offload-arch=gfx950: HIP kernels + the C ABI of include/fsmi355.h  (the product)
  host/libfsinputs.so  g++ + GMP: host-side input builders of include/fs_inputs.h (view / orbit / LA / BLA)

`-ffp-contract=off` is part of the numerical contract (see csrc/hdr_math.hpp): the parity target is the
reference's CPU build, which has no FMA.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB_RENDER = os.path.join(CSRC, "libfsmi355.so")
LIB_INPUTS = os.path.join(HOST, "libfsinputs.so")

GMP_PREFIX = os.environ.get("FS_GMP_PREFIX", "/opt/conda")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def build_render(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("kernels.hip", "kernels_2x32.hip", "kernels_scaled.hip", "kernels_tables.hip", "kernels_direct_lp.hip", "kernels_plain.hip", "kernels_decompress.hip", "renderer.cpp", "kernels.h", "kernel_common.hpp",
                                            "hdr_math.hpp", "df32_math.hpp")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("fsmi355.h", "fs_layout.h")]
    if not force and _newer(LIB_RENDER, srcs):
        return LIB_RENDER
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # FS_PROFILE_CYCLES=1: the instrumented (step-counting) kernel variants also report shader-clock cycles per phase
    # (tools/cycle_probe.py); never set for the product build
    extra = ["-DFS_PROFILE_CYCLES"] if os.environ.get("FS_PROFILE_CYCLES") == "1" else []
    if os.environ.get("FS_SCALED_CHUNK"):  # tuning experiments only
        extra.append("-DFS_SCALED_CHUNK=" + str(int(os.environ["FS_SCALED_CHUNK"])))
    _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", *extra,
          "-o", LIB_RENDER, os.path.join(CSRC, "kernels.hip"), os.path.join(CSRC, "kernels_2x32.hip"),
          os.path.join(CSRC, "kernels_scaled.hip"), os.path.join(CSRC, "kernels_tables.hip"),
          os.path.join(CSRC, "kernels_direct_lp.hip"), os.path.join(CSRC, "kernels_plain.hip"), os.path.join(CSRC, "kernels_decompress.hip"),
          os.path.join(CSRC, "renderer.cpp")])
    return LIB_RENDER


def build_inputs(force=False):
    srcs = [os.path.join(HOST, "refinputs.cpp"), os.path.join(CSRC, "hdr_math.hpp"),
            os.path.join(ROOT, "include", "fs_inputs.h"), os.path.join(ROOT, "include", "fs_layout.h")]
    if not force and _newer(LIB_INPUTS, srcs):
        return LIB_INPUTS
    _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
          "-I" + os.path.join(GMP_PREFIX, "include"), "-o", LIB_INPUTS, os.path.join(HOST, "refinputs.cpp"),
          "-L" + os.path.join(GMP_PREFIX, "lib"), "-lgmp", "-Wl,-rpath," + os.path.join(GMP_PREFIX, "lib")])
    return LIB_INPUTS


def build_all(force=False):
    return build_render(force), build_inputs(force)


if __name__ == "__main__":
    print(build_all(force=True))
