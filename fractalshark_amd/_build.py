"""Build recipes for the native libraries (in-tree, so the .so files travel to the GPU box).

  csrc/libfsmi355.so   hipcc --offload-arch=gfx950: HIP kernels + the C ABI of include/fsmi355.h  (the product)
  host/libfsinputs.so  g++ + GMP: host-side input builders of include/fs_inputs.h (view / orbit / LA / BLA)

`-ffp-contract=off` is part of the numerical contract (see csrc/hdr_math.hpp): the parity target is the
reference's CPU build, which has no FMA.

Up-to-date checks compare a CONTENT hash of the sources + flags with a stamp written next to each output (file
times do not survive the snapshot that carries the tree to the GPU box; a stale time there would start a compiler
under a profiler).  Translation units are compiled to objects in parallel and only the changed ones are rebuilt.
"""
import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
OBJ = os.path.join(CSRC, "build")
LIB_RENDER = os.path.join(CSRC, "libfsmi355.so")
LIB_INPUTS = os.path.join(HOST, "libfsinputs.so")

GMP_PREFIX = os.environ.get("FS_GMP_PREFIX", "/opt/conda")


def _run(cmd):
    # compilers never inherit a profiler's preload (rocprofv3 injects a library that initialises the GPU; exec'ing
    # clang / cc1plus / ld from such a process tree is what the GPU pool forbids)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF"))}
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    if p.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def _digest(paths, flags):
    h = hashlib.sha256()
    h.update("\0".join(flags).encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(target, digest):
    try:
        return os.path.exists(target) and open(target + ".stamp").read().strip() == digest
    except OSError:
        return False


def _write_stamp(target, digest):
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


def _render_units():
    """Translation units of libfsmi355.so: every csrc/*.hip plus the host side of the C ABI."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(CSRC, "renderer.cpp"),
                                                             os.path.join(CSRC, "group.cpp")]


def _render_headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                  [os.path.join(ROOT, "include", f) for f in ("fsmi355.h", "fsmi355_internal.h", "fs_layout.h")])


def _render_flags():
    # FS_PROFILE_CYCLES=1: the instrumented (step-counting) kernel variants also report shader-clock cycles per phase
    # (tools/cycle_probe.py); never set for the product build
    extra = ["-DFS_PROFILE_CYCLES"] if os.environ.get("FS_PROFILE_CYCLES") == "1" else []
    if os.environ.get("FS_VERIFY_BLOCK_BOUND") == "1":  # tools/block_bound_check.py: the untested loop off, violations counted
        extra.append("-DFS_VERIFY_BLOCK_BOUND")
    if os.environ.get("FS_BACKOFF_CAP"):  # A/B: back-off cap of the scaled-run attempts (scaled_runs.hpp; 0 = off)
        extra.append("-DFS_BACKOFF_CAP=%d" % int(os.environ["FS_BACKOFF_CAP"]))
    if os.environ.get("FS_TRACE_WAVES") == "1":  # tools/wave_trace.py: per-wave start / end / SIMD records
        extra.append("-DFS_TRACE_WAVES")
    if os.environ.get("FS_VERIFY_FLOOR") == "1":  # tools/floor_check.py: trips whose untested first state is below the every-state floor
        extra.append("-DFS_VERIFY_FLOOR")
    if os.environ.get("FS_BLA_FAST_PROBE") == "1":  # tools/bla_fast_check.py: how often the hand-written BLA loop is left (statistics words 20..23)
        extra.append("-DFS_BLA_FAST_PROBE")
    if os.environ.get("FS_FD_LANE_BOUND") == "1":  # A/B: round 4's per-lane block test in C3's untested loop
        extra.append("-DFS_FD_LANE_BOUND")
    if os.environ.get("FS_FD16_SERIAL") == "1":  # A/B: round 4's 16-step body (wait right behind the request) instead of the pipelined one
        extra.append("-DFS_FD16_SERIAL")
    if os.environ.get("FS_2X32_PROBE") == "1":  # counts the 2x32 perturbation loop's literal steps (statistics word 12)
        extra.append("-DFS_2X32_PROBE")
    for name in ("FS_FL_EVERY", "FS_FL_SHIFT", "FS_FL_FLOOR_EXP", "FS_HOT_RUN_STEPS", "FS_PO_CHUNK", "FS_AT_CYCLE_CHUNK", "FS_HOT_AFTER_FAIL"):  # A/B: form and scale of the scaled runs' floor tests (kernels.hip)
        if os.environ.get(name):
            extra.append("-D%s=%d" % (name, int(os.environ[name])))
    if os.environ.get("FS_SCALED_CHUNK"):  # tuning experiments only
        extra.append("-DFS_SCALED_CHUNK=" + str(int(os.environ["FS_SCALED_CHUNK"])))
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *extra]


# Per-unit flags.  kernels_scaled.hip: the scalar float expressions of the reference's scaled kernel are written out
# operation by operation; the SLP vectorizer pairs them into packed instructions with register shuffles around them, which
# costs more than the two-operand scalar forms it replaces (the pairs that pay are written as float2 in the source).
_UNIT_FLAGS = {"kernels_scaled.hip": ["-fno-slp-vectorize"]}


# Units whose DEVICE code goes through tools/asm_peephole.py between the compiler and the assembler (v_cndmask_b32_e32 -> _e64).
# EMPTY: back to back the 32-bit encoding of the select issues five times slower than the 64-bit one on gfx950
# (profiles/r06_valu_issue_rates_f64.jsonl), but in a kernel -- k_lav2_hdr64, 838 selects rewritten, same box -- the frame time did not
# move (33.33 against 33.28 ms, profiles/r06_c4_hdr64_kernel_ab_same_box.jsonl): the selects of real code are not back to back.  The
# step stays for A/B builds (FS_PEEPHOLE_UNITS=kernels_x.hip,... in the environment of tools/build_variant.py).
PEEPHOLE_TOOL = os.path.join(ROOT, "tools", "asm_peephole.py")
_PEEPHOLE_UNITS = set(u for u in os.environ.get("FS_PEEPHOLE_UNITS", "").split(",") if u)
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _peephole_state():
    return "peephole:" + (",".join(sorted(_PEEPHOLE_UNITS)) if os.environ.get("FS_PEEPHOLE", "1") != "0" else "off")


def _peephole_on(src):
    return src.endswith(".hip") and os.path.basename(src) in _PEEPHOLE_UNITS and os.environ.get("FS_PEEPHOLE", "1") != "0"


def compile_one(hipcc, src, obj, flags, peephole):
    """One translation unit -> object.  Plain: hipcc -c.  With the peephole: the driver's own steps taken apart -- device code to
    assembly, the rewrite, assembler, lld (code object), clang-offload-bundler (fat binary), then the host pass with that binary."""
    # renderer.cpp / group.cpp are host-only C++ that include HIP runtime headers: compiled by hipcc as HIP so
    # that <hip/hip_runtime.h> types (float4, hipStream_t) match the kernels' launchers
    lang = [] if src.endswith(".hip") else ["-x", "hip"]
    if not peephole:
        _run([hipcc, *flags, "-c", *lang, src, "-o", obj])
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("fs_asm_peephole", PEEPHOLE_TOOL)
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    base = obj[:-2] if obj.endswith(".o") else obj
    asm, asm2, dev, hsaco, fb = base + ".dev.s", base + ".dev.pp.s", base + ".dev.o", base + ".hsaco", base + ".hipfb"
    _run([hipcc, *flags, "--cuda-device-only", "-S", *lang, src, "-o", asm])
    text, n = tool.rewrite(open(asm).read())
    open(asm2, "w").write(text)
    _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm2, "-o", dev])
    _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev])
    _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco,
          "-output=" + fb])
    _run([hipcc, *flags, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", *lang, src, "-o", obj])
    for f in (asm, dev, hsaco, fb):  # (the rewritten assembly stays next to the object: what was assembled can be read)
        try:
            os.remove(f)
        except OSError:
            pass


def _inputs_sources():
    return [os.path.join(HOST, "refinputs.cpp"), os.path.join(CSRC, "hdr_math.hpp"), os.path.join(CSRC, "la_math.hpp"),
            os.path.join(CSRC, "df32_math.hpp"), os.path.join(CSRC, "bla_math.hpp"),
            os.path.join(ROOT, "include", "fs_inputs.h"), os.path.join(ROOT, "include", "fs_layout.h")]


_INPUTS_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def up_to_date():
    """True when both libraries exist and were built from the current sources (no compiler is started)."""
    units = [u for u in _render_units() if os.path.exists(u)]
    return (_stamp_ok(LIB_RENDER, _digest(units + _render_headers() + [PEEPHOLE_TOOL],
                                          _render_flags() + [repr(sorted(_UNIT_FLAGS.items())), _peephole_state()])) and
            _stamp_ok(LIB_INPUTS, _digest(_inputs_sources(), _INPUTS_FLAGS)))


def build_render(force=False):
    units = [u for u in _render_units() if os.path.exists(u)]
    headers = _render_headers()
    flags = _render_flags()
    digest = _digest(units + headers + [PEEPHOLE_TOOL], flags + [repr(sorted(_UNIT_FLAGS.items())), _peephole_state()])
    if not force and _stamp_ok(LIB_RENDER, digest):
        return LIB_RENDER
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    hdr_digest = _digest(headers, flags)

    def compile_unit(src):
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        unit_flags = _UNIT_FLAGS.get(os.path.basename(src), [])
        peep = _peephole_on(src)
        d = _digest([src] + ([PEEPHOLE_TOOL] if peep else []), [hdr_digest, *unit_flags, "peephole" if peep else ""])
        if force or not _stamp_ok(obj, d):
            compile_one(hipcc, src, obj, [*flags, *unit_flags], peep)
            _write_stamp(obj, d)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_unit, units))
    # RCCL (the multi-GPU gather behind fs_group_*) is resolved at run time with dlopen, so the library loads on hosts
    # without it; -ldl only
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_RENDER, *objs, "-ldl", "-lpthread"])
    _write_stamp(LIB_RENDER, digest)
    return LIB_RENDER


def build_inputs(force=False):
    srcs = _inputs_sources()
    digest = _digest(srcs, _INPUTS_FLAGS)
    if not force and _stamp_ok(LIB_INPUTS, digest):
        return LIB_INPUTS
    _run(["g++", *_INPUTS_FLAGS, "-I" + os.path.join(GMP_PREFIX, "include"), "-o", LIB_INPUTS,
          os.path.join(HOST, "refinputs.cpp"), "-L" + os.path.join(GMP_PREFIX, "lib"), "-lgmp",
          "-Wl,-rpath," + os.path.join(GMP_PREFIX, "lib")])
    _write_stamp(LIB_INPUTS, digest)
    return LIB_INPUTS


def build_all(force=False):
    return build_render(force), build_inputs(force)


if __name__ == "__main__":
    print(build_all(force=True))


def status2_test_variant():
    """build/ab/libfsmi355_h64st2.so: the library with k_lav2_hdr64's hand-written statements taking their rarest exit (status 2) on
    EVERY step (-DFS_H64_ASM_TINY=1e300) -- what tests/test_gpu_hdr64_statement_exits.py renders with.  Built by tools/build_variant.py
    when its content stamp does not match the sources; returns the path."""
    import sys
    defs = ["-DFS_H64_ASM_TINY=1e300"]
    lib = os.path.join(ROOT, "build", "ab", "libfsmi355_h64st2.so")
    stamp = lib + ".stamp"  # (by content, as the product's own stamp: file times mean nothing on a freshly copied tree)
    digest = _digest(_render_units() + _render_headers(), _render_flags() + defs)
    if not os.path.exists(lib) or not os.path.exists(stamp) or open(stamp).read().strip() != digest:
        env = {k: v for k, v in os.environ.items() if k != "FSMI355_LIB"}
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "h64st2", "kernels_hdr64.hip", *defs], cwd=ROOT,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError("build of the status-2 test variant failed:\n" + p.stdout[-3000:])
        with open(stamp, "w") as f:
            f.write(digest + "\n")
    return lib
