"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
PIN = os.path.join(ORACLE_DIR, "_ref", "libpngpin.so")

vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
_lib = None
_pin = None


def build():
    """make the oracle (and, where /root/reference exists, the golden-CRC pin).  Up-to-date checks are content hashes
    (fractalshark_amd._build), not file times; with FS_NO_BUILD=1 (profiler runs) nothing is ever compiled."""
    import glob

    from fractalshark_amd import _build
    no_build = os.environ.get("FS_NO_BUILD") == "1"
    srcs = sorted(glob.glob(os.path.join(ORACLE_DIR, "*.cpp")) + glob.glob(os.path.join(ORACLE_DIR, "*.hpp")) +
                  [os.path.join(ORACLE_DIR, "Makefile"), os.path.join(ROOT, "include", "fs_layout.h")])
    lib_srcs = [s for s in srcs if not s.endswith("png_pin.cpp")]
    d = _build._digest(lib_srcs, ["oracle"])
    if not _build._stamp_ok(LIB, d):
        if no_build:
            raise RuntimeError("oracle/liboracle.so is missing or stale and FS_NO_BUILD=1 forbids compiling")
        _build._run(["make", "-C", ORACLE_DIR, "-B", "all"])
        _build._write_stamp(LIB, d)
        try:  # what built it, for bench.py's cpu_baseline line (read back there: a timed process spawns nothing)
            ver = _build._run(["g++", "--version"]).splitlines()[0].strip()
        except Exception:
            ver = "g++ (version unknown)"
        with open(LIB + ".compiler", "w") as f:
            f.write(ver + "\n")
    # the golden-CRC pin needs the reference's WPngImage/lodepng sources: only buildable where /root/reference is
    pin_srcs = [os.path.join(ORACLE_DIR, "png_pin.cpp"), os.path.join(ORACLE_DIR, "Makefile")]
    d = _build._digest(pin_srcs, ["pin"])
    if not _build._stamp_ok(PIN, d) and not no_build and os.path.isdir("/root/reference/FractalSharkLib/WPngImage"):
        _build._run(["make", "-C", ORACLE_DIR, "-B", "_ref"])
        _build._write_stamp(PIN, d)


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB)
        l.orc_direct_f64.restype = None
        l.orc_direct_f64.argtypes = [u32, u32, u32, u32, vp, u32, vp, u32, C.c_int]
        l.orc_bla_hdr32.restype = None
        l.orc_bla_hdr32.argtypes = [u32, u32, u32, u32, vp, u64, vp, u32, vp, vp, i32, i32, vp, u32, C.c_int]
        l.orc_lav2_hdr32.restype = None
        l.orc_lav2_hdr32.argtypes = [u32, u32, u32, u32, vp, u64, u64, vp, u32, vp, u32, C.c_int, C.c_int, vp, vp,
                                     u32, C.c_int, C.c_int, vp, u32, C.c_int, vp]
        for name in ("orc_bla_hdr64", "orc_lav2_hdr64", "orc_direct_hdr32", "orc_direct_hdr64"):
            getattr(l, name).restype = None
        l.orc_bla_hdr64.argtypes = l.orc_bla_hdr32.argtypes
        l.orc_lav2_hdr64.argtypes = l.orc_lav2_hdr32.argtypes
        l.orc_direct_hdr32.argtypes = [u32, u32, u32, u32, vp, u32, vp, u32, C.c_int]
        l.orc_direct_hdr64.argtypes = l.orc_direct_hdr32.argtypes
        for name in ("orc_lav2_hdr32_u64", "orc_lav2_hdr64_u64"):
            getattr(l, name).restype = None
            getattr(l, name).argtypes = [u32, u32, u32, u32, vp, u64, u64, vp, u32, vp, u32, C.c_int, C.c_int, vp, vp,
                                         u64, C.c_int, C.c_int, vp, u32, C.c_int, vp]
        l.orc_bla_f64.restype = None
        l.orc_bla_f64.argtypes = [u32, u32, u32, u32, vp, u64, vp, u32, vp, vp, i32, i32, vp, u32, C.c_int]
        l.orc_gpu_lav2_2x32.restype = None
        l.orc_gpu_lav2_2x32.argtypes = [vp, u32, u32, u32, u32, vp, u32, vp, vp, u32, C.c_int, C.c_int, vp, vp, u32,
                                        C.c_int, C.c_int, vp]
        l.orc_gpu_lav2_plain.restype = None
        l.orc_gpu_lav2_plain.argtypes = [C.c_int, vp, u32, u32, u32, u32, vp, u32, vp, vp, u32, C.c_int, C.c_int, vp, vp,
                                         u32, C.c_int, C.c_int, vp]
        for name in ("orc_df_add", "orc_df_sub", "orc_df_mul"):
            getattr(l, name).restype = None
            getattr(l, name).argtypes = [vp, vp, vp]
        l.orc_h2_reduce.restype = None
        l.orc_h2_reduce.argtypes = [vp]
        l.orc_h2_add.restype = None
        l.orc_h2_add.argtypes = [vp, vp, C.c_int, vp]
        l.orc_c2_reduce.restype = None
        l.orc_c2_reduce.argtypes = [vp]
        for name in ("orc_gpu_direct_1x32", "orc_gpu_direct_2x32"):
            getattr(l, name).restype = None
            getattr(l, name).argtypes = [vp, u32, u32, u32, u32, u32, vp, u32, C.c_int]
        for name in ("orc_gpu_direct_2x64", "orc_gpu_direct_4x32", "orc_gpu_direct_4x64"):
            getattr(l, name).restype = None
            getattr(l, name).argtypes = [vp, u32, u32, u32, u32, u32, vp, u32]
        l.orc_gpu_scaled_hdr32.restype = None
        l.orc_gpu_scaled_hdr32.argtypes = [vp, u32, u32, u32, u32, vp, vp, u32, vp, u32, C.c_float, C.c_int, vp]
        l.orc_gpu_scaled_f64.restype = None
        l.orc_gpu_scaled_f64.argtypes = [vp, u32, u32, u32, u32, vp, vp, u32, vp, u32, C.c_float, C.c_int, vp]
        l.orc_set_row_step.restype = None
        l.orc_set_row_step.argtypes = [u32]
        _lib = l
    return _lib


def compiler_and_flags():
    """Compiler and flags the oracle library was built with (oracle/Makefile: the reference's Linux release flags, -O3 and no
    -march, build_linux.sh:20-23), for the cpu_baseline line."""
    flags = "-O3 -std=c++17 -ffp-contract=off -fPIC -pthread"
    try:
        for ln in open(os.path.join(ORACLE_DIR, "Makefile")):
            if ln.startswith("CXXFLAGS"):
                flags = ln.split("=", 1)[1].strip()
    except OSError:
        pass
    try:
        ver = open(LIB + ".compiler").read().strip()
    except OSError:
        ver = "g++"
    return "%s %s" % (ver, flags)


def set_row_step(step):
    """Render only rows y0, y0+step, ... (evenly spread sample for the bounded cpu_baseline timing)."""
    lib().orc_set_row_step(step)


def workload_rows(inp, y0, y1, threads=8, stage_test=None, row_step=1):
    """Rows y0, y0 + row_step, ... < y1 of a bench workload's frame (inp = bench.make_inputs(...)) as the oracle renders them:
    the ONE dispatch from workload to oracle function, shared by bench.py's cpu_baseline leg and
    tests/golden/make_frame_crcs.py.  stage_test None follows inp["parity"] (cpu = 0: the literal CPU function)."""
    if stage_test is None:
        stage_test = 0 if inp["parity"] == "cpu" else 1
    view, orbit, aa, n = inp["view"], inp["orbit"], inp["AA"], inp["n_iter"]
    set_row_step(row_step)
    try:
        if inp.get("is_direct"):
            # (Fractal.cpp:2096-2206; direct_f64 renders every row of [y0, y1): a one-second frame, no row step)
            return direct_f64(view, aa=aa, rows=(y0, y1), threads=threads)
        if inp["is2x32"]:
            return gpu_lav2_2x32(view, inp["orbit2"], inp["la2"], aa=aa, rows=(y0, y1), threads=threads, n_iterations=n)
        if inp["is_lav2"]:
            return lav2_hdr32(view, orbit, inp["la"], aa=aa, rows=(y0, y1), threads=threads, stage_test=stage_test,
                              n_iterations=n)
        if inp["is_scaled"]:
            return gpu_scaled_hdr32(view, orbit, aa=aa, rows=(y0, y1), threads=threads, n_iterations=n)
        return bla_hdr32(view, orbit, inp["bla"], aa=aa, rows=(y0, y1), threads=threads, n_iterations=n)
    finally:
        set_row_step(1)


def pin_lib():
    global _pin
    if _pin is None:
        build()
        if not os.path.exists(PIN):
            return None
        p = C.CDLL(PIN)
        p.pin_png_crc64.restype = u64
        p.pin_png_crc64.argtypes = [vp, u32, u32, u32, u32, u64, u64, C.c_char_p]
        p.pin_png_crc64_rgba16.restype = u64
        p.pin_png_crc64_rgba16.argtypes = [vp, u32, u32, u32]
        p.pin_default_palette.restype = u32
        p.pin_default_palette.argtypes = [C.c_int, vp, u32]
        _pin = p
    return _pin


def rounded_width(w):
    return (w + 15) // 16 * 16


def new_buffer(w, h):
    return np.zeros(((h + 7) // 8 * 8, rounded_width(w)), np.uint32)


def direct_f64(view, aa=1, rows=None, threads=8):
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_direct_f64(aa)
    y0, y1 = rows if rows else (0, h)
    lib().orc_direct_f64(w, h, y0, y1, co.ctypes.data, view.num_iterations, out.ctypes.data, out.shape[1], threads)
    return out


def direct_hdr(view, is64, aa=1, rows=None, threads=8):
    """CalcCpuHDR<u32,HDRFloat<F>,F> (CpuHDR32 / CpuHDR64)."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_direct_hdr(is64, aa)
    y0, y1 = rows if rows else (0, h)
    fn = lib().orc_direct_hdr64 if is64 else lib().orc_direct_hdr32
    fn(w, h, y0, y1, co.ctypes.data, view.num_iterations, out.ctypes.data, out.shape[1], threads)
    return out


def bla_hdr32(view, orbit, bla=None, aa=1, rows=None, threads=8, n_iterations=None):
    """CalcCpuPerturbationFractalBLA<u32,HDRFloat<F>,F> (F follows the orbit); bla=None suppresses the lookup
    (C2 target)."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_perturb(orbit, aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    if orbit.is64:
        fn = lib().orc_bla_hdr64
        if bla is None:
            fn(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, n, None, None, 0, 0, out.ctypes.data,
               out.shape[1], threads)
        else:
            fn(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, n, bla.level_ptrs, bla.level_sizes,
               bla.num_levels, bla.lm2, out.ctypes.data, out.shape[1], threads)
        return out
    if bla is None:
        lib().orc_bla_hdr32(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, n, None, None, 0, 0,
                            out.ctypes.data, out.shape[1], threads)
    else:
        lib().orc_bla_hdr32(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, n, bla.level_ptrs,
                            bla.level_sizes, bla.num_levels, bla.lm2, out.ctypes.data, out.shape[1], threads)
    return out


def lav2_hdr32(view, orbit, la, aa=1, rows=None, threads=8, stage_test=0, mode=0, n_iterations=None, stats=False):
    """CalcCpuPerturbationFractalLAV2<u32,float,Disable>. stage_test 0 = literal CPU, 1 = GPU direction."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_perturb(orbit, aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    st = (u64 * 4)()
    fn = lib().orc_lav2_hdr64 if orbit.is64 else lib().orc_lav2_hdr32
    fn(w, h, y0, y1, orbit.data_ptr, orbit.count, orbit.period, la.las_ptr, la.count,
                         la.stages_ptr, la.stage_count, 1 if la.is_valid else 0, 1 if la.use_at else 0,
                         C.addressof(la.at), co.ctypes.data, n, stage_test, mode, out.ctypes.data, out.shape[1],
                         threads, st)
    if stats:
        return out, {"at_iterations": st[0], "la_steps": st[1], "perturb_steps": st[2], "pixels": st[3]}
    return out


def lav2_u64(view, orbit, la, n_iterations, aa=1, rows=None, threads=8, stage_test=0, mode=0):
    """CalcCpuPerturbationFractalLAV2<uint64_t, ...>: 64-bit counters and a uint64 buffer (iteration caps >= 2^32)."""
    w, h = view.width * aa, view.height * aa
    out = np.zeros(((h + 7) // 8 * 8, rounded_width(w)), np.uint64)
    co = view.coords_perturb(orbit, aa)
    y0, y1 = rows if rows else (0, h)
    st = (u64 * 4)()
    fn = lib().orc_lav2_hdr64_u64 if orbit.is64 else lib().orc_lav2_hdr32_u64
    fn(w, h, y0, y1, orbit.data_ptr, orbit.count, orbit.period, la.las_ptr, la.count, la.stages_ptr, la.stage_count,
       1 if la.is_valid else 0, 1 if la.use_at else 0, C.addressof(la.at), co.ctypes.data, int(n_iterations), stage_test,
       mode, out.ctypes.data, out.shape[1], threads, st)
    return out


def decompress_hdr2x32(orbit2):
    """The full orbit the reference's GPU rebuilds from a SimpleCompression HDRFloat<CudaDblflt> waypoint list
    (GetCompressedComplexSeq in 2x32 arithmetic, oracle/gpu_ref_2x32.cpp).  orbit2: a compressed inputs.Orbit2x32."""
    from fractalshark_amd.inputs import ORBIT_2X32_DTYPE
    wp, low = orbit2.waypoints(), orbit2.orbit_low()
    out = np.zeros(orbit2.count, ORBIT_2X32_DTYPE)
    fn = lib().orc_decompress_hdr2x32
    fn.restype, fn.argtypes = None, [C.c_void_p, u64, u64, C.c_void_p, C.c_void_p]
    fn(wp.ctypes.data, len(wp), orbit2.count, low.ctypes.data, out.ctypes.data)
    return out


def decompress_p2x32(pin):
    """The same for CudaDblflt (Gpu2x32PerturbedRCLAv2*).  pin: a compressed inputs.PlainInputs of kind "2x32"."""
    wp, low = pin.waypoints(), pin.orbit_low()
    out = np.zeros(pin.count, pin._orbit.dtype)
    fn = lib().orc_decompress_p2x32
    fn.restype, fn.argtypes = None, [C.c_void_p, u64, u64, C.c_void_p, C.c_void_p]
    fn(wp.ctypes.data, len(wp), pin.count, low.ctypes.data, out.ctypes.data)
    return out


def gpu_lav2_2x32(view, orbit2, la2, aa=1, rows=None, threads=8, mode=0, n_iterations=None, stats=False,
                  orbit_entries=None):
    """Restated CUDA kernel mandel_1xHDR_float_perturb_lav2<.., HDRFloat<CudaDblflt>, ..> (oracle/gpu_ref_2x32.cpp).
    orbit2: inputs.Orbit2x32; la2: inputs.LATable2x32 or None (mode 1 = perturbation only).  orbit_entries: the
    uncompressed entries to use instead of orbit2's own (decompress_hdr2x32 for a SimpleCompression orbit)."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_perturb_2x32(orbit2, aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    st = (u64 * 3)()
    if orbit_entries is not None:
        class _O:
            data_ptr, count = orbit_entries.ctypes.data, len(orbit_entries)
        orbit2 = _O
    if la2 is not None:
        lib().orc_gpu_lav2_2x32(out.ctypes.data, out.shape[1], w, y0, y1, orbit2.data_ptr, orbit2.count, la2.las_ptr,
                                la2.stages_ptr, la2.stage_count, 1 if la2.is_valid else 0, 1 if la2.use_at else 0,
                                C.addressof(la2.at), co.ctypes.data, n, mode, threads, st)
    else:
        lib().orc_gpu_lav2_2x32(out.ctypes.data, out.shape[1], w, y0, y1, orbit2.data_ptr, orbit2.count, None, None, 0,
                                0, 0, None, co.ctypes.data, n, mode, threads, st)
    if stats:
        return out, {"at_iterations": st[0], "la_steps": st[1], "perturb_steps": st[2]}
    return out


def gpu_lav2_plain(view, pin, aa=1, rows=None, threads=8, mode=0, n_iterations=None, stats=False, orbit_entries=None):
    """Restated CUDA kernel mandel_1xHDR_float_perturb_lav2<.., T, T, ..> for T = float / double / CudaDblflt
    (oracle/gpu_ref_plain.cpp).  pin: inputs.PlainInputs (its kind selects T); mode 0 Full, 1 PO, 2 LAO."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    st = (u64 * 3)()
    kind = {"f32": 0, "f64": 1, "2x32": 2}[pin.kind]
    orbit_ptr = pin.orbit_ptr if orbit_entries is None else orbit_entries.ctypes.data
    lib().orc_gpu_lav2_plain(kind, out.ctypes.data, out.shape[1], w, y0, y1, orbit_ptr, pin.count, pin.las_ptr,
                             pin.stages_ptr, pin.stage_count, 1 if pin.is_valid else 0, 1 if pin.use_at else 0,
                             pin.at_ptr, pin.coords_ptr, n, mode, threads, st)
    if stats:
        return out, {"at_iterations": st[0], "la_steps": st[1], "perturb_steps": st[2]}
    return out


def gpu_scaled_hdr32(view, orbit, aa=1, rows=None, threads=8, n_iterations=None, stats=False):
    """Restated CUDA kernel mandel_1x_float_perturb_scaled<.., HDRFloat<float>> (oracle/cpu_ref.cpp); orbit: hdr32."""
    import math
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_perturb(orbit, aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    st = (u64 * 3)()
    w2 = float(np.float32(math.exp(math.log(float(np.float32(1e30))) / 2.0)))
    lib().orc_gpu_scaled_hdr32(out.ctypes.data, out.shape[1], w, y0, y1, orbit.bad_data_ptr, orbit.bad_f32_data_ptr,
                               orbit.count, co.ctypes.data, n, w2, threads, st)
    if stats:
        return out, {"rescales": st[0], "full_steps": st[1], "float_steps": st[2]}
    return out


def gpu_scaled_f64(view, orbit, aa=1, rows=None, threads=8, n_iterations=None, stats=False):
    """Restated CUDA kernel mandel_1x_float_perturb_scaled<.., double> (oracle/cpu_ref.cpp); orbit: inputs.OrbitF64."""
    import math
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = orbit.coords(aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    st = (u64 * 3)()
    w2 = float(np.float32(math.exp(math.log(float(np.float32(1e30))) / 2.0)))
    lib().orc_gpu_scaled_f64(out.ctypes.data, out.shape[1], w, y0, y1, orbit.bad_data_ptr, orbit.bad_f32_data_ptr,
                             orbit.count, co.ctypes.data, n, w2, threads, st)
    if stats:
        return out, {"rescales": st[0], "full_steps": st[1], "float_steps": st[2]}
    return out


def gpu_direct_lp(view, kind, iteration_precision=1, aa=1, rows=None, n_iterations=None):
    """Restated CUDA direct kernels without a CPU twin (oracle/gpu_ref_lp.cpp).  kind: "1x32" | "2x32" | "2x64", and
    "4x32" | "4x64" (oracle/gpu_ref_qd.cpp)."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = view.coords_direct_lp(kind, aa)
    y0, y1 = rows if rows else (0, h)
    n = view.num_iterations if n_iterations is None else n_iterations
    if kind in ("2x64", "4x32", "4x64"):
        fn = {"2x64": lib().orc_gpu_direct_2x64, "4x32": lib().orc_gpu_direct_4x32, "4x64": lib().orc_gpu_direct_4x64}[kind]
        fn(out.ctypes.data, out.shape[1], w, h, y0, y1, co.ctypes.data, n)
    else:
        fn = lib().orc_gpu_direct_1x32 if kind == "1x32" else lib().orc_gpu_direct_2x32
        fn(out.ctypes.data, out.shape[1], w, h, y0, y1, co.ctypes.data, n, iteration_precision)
    return out


def bla_f64(view, orbit, use_bla=True, aa=1, rows=None, threads=8):
    """CalcCpuPerturbationFractalBLA<u32,double,double> (Cpu64PerturbedBLA); orbit: inputs.OrbitF64."""
    w, h = view.width * aa, view.height * aa
    out = new_buffer(w, h)
    co = orbit.coords(aa)
    y0, y1 = rows if rows else (0, h)
    if use_bla:
        lib().orc_bla_f64(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, view.num_iterations,
                          orbit.level_ptrs, orbit.level_sizes, orbit.num_levels, orbit.lm2, out.ctypes.data,
                          out.shape[1], threads)
    else:
        lib().orc_bla_f64(w, h, y0, y1, orbit.data_ptr, orbit.count, co.ctypes.data, view.num_iterations, None, None,
                          0, 0, out.ctypes.data, out.shape[1], threads)
    return out


def png_crc64(iters, width, height, aa, num_iterations, save_path=None):
    """CRC-64 of the PNG bytes the reference's headless render writes for this iteration buffer, or None when the
    pin library (needs /root/reference at build time) is unavailable."""
    p = pin_lib()
    if p is None:
        return None
    it = np.ascontiguousarray(iters, np.uint32)
    # GetMaxIterations<uint32_t>() = INT32_MAX - 1 (Fractal.h:118-123)
    crc = p.pin_png_crc64(it.ctypes.data, it.shape[1], width, height, aa, num_iterations, 2 ** 31 - 2,
                          save_path.encode() if save_path else None)
    return "%016x" % crc


def png_crc64_rgba16(colors, width, height):
    """CRC-64 of the PNG the reference writes for an RGBA16 colour buffer (uint16[rows, width, 4], already antialiased
    and palette-mapped: GPURenderer::RenderCurrent's Color16 output), or None without the pin library."""
    p = pin_lib()
    if p is None:
        return None
    c = np.ascontiguousarray(colors, np.uint16)
    return "%016x" % p.pin_png_crc64_rgba16(c.ctypes.data, width, width, height)


def default_palette(depth=8):
    """Default palette as uint16[N,4]; restated in oracle/png_pin.cpp, falls back to a local restatement."""
    p = pin_lib()
    n = 7 << depth
    out = np.zeros((n, 4), np.uint16)
    if p is not None:
        p.pin_default_palette(depth, out.ctypes.data, n)
        return out
    # FractalPalette.cpp:27-46,147-174
    pal = []
    cur = (0, 0, 0)
    mv = 65535
    for tgt in ((mv, 0, 0), (mv, mv, 0), (0, mv, 0), (0, mv, mv), (0, 0, mv), (mv, 0, mv), (0, 0, 0)):
        length = 1 << depth
        d = [(tgt[k] - cur[k]) / length for k in range(3)]
        for i in range(length):
            pal.append(tuple(int(cur[k] + d[k] * (i + 1)) & 0xFFFF for k in range(3)) + (0,))
        cur = pal[-1][:3]
    return np.array(pal, np.uint16)
