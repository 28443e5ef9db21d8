"""Host-side inputs of the per-pixel pass: view geometry, reference orbit, LAv2 table, BLA table.

Thin Python handles over libfsinputs.so (include/fs_inputs.h).  Inside FractalShark these objects already
exist (PointZoomBBConverter, PerturbationResults, LAReference, BLAS); here they are produced stand-alone so
the renderer can be driven from bench.py and the tests.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _capi

_VIEWS = None

ORBIT_HDR32_DTYPE = np.dtype([("mx", "<f4"), ("ex", "<i4"), ("ey", "<i4"), ("my", "<f4")])
LA_HDR32_DTYPE = np.dtype([
    ("Ref_re", "<f4"), ("Ref_im", "<f4"), ("Ref_e", "<i4"),
    ("Z_re", "<f4"), ("Z_im", "<f4"), ("Z_e", "<i4"),
    ("C_re", "<f4"), ("C_im", "<f4"), ("C_e", "<i4"),
    ("LAThreshold_m", "<f4"), ("LAThreshold_e", "<i4"),
    ("LAThresholdC_m", "<f4"), ("LAThresholdC_e", "<i4"),
    ("MinMag_m", "<f4"), ("MinMag_e", "<i4"),
    ("StepLength", "<u4"), ("NextStageLAIndex", "<u4")])
BLA_HDR32_DTYPE = np.dtype([("r2_m", "<f4"), ("r2_e", "<i4"), ("Ax_m", "<f4"), ("Ax_e", "<i4"),
                            ("Ay_m", "<f4"), ("Ay_e", "<i4"), ("Bx_m", "<f4"), ("Bx_e", "<i4"),
                            ("By_m", "<f4"), ("By_e", "<i4"), ("l", "<i4")])
ORBIT_HDR64_DTYPE = np.dtype([("mx", "<f8"), ("ex", "<i4"), ("pad0", "<i4"), ("ey", "<i4"), ("pad1", "<i4"),
                              ("my", "<f8")])
REAL_HDR32 = np.dtype([("m", "<f4"), ("e", "<i4")])
REAL_HDR64 = np.dtype([("m", "<f8"), ("e", "<i4"), ("pad_", "<i4")])
REAL_2X32 = np.dtype([("head", "<f4"), ("tail", "<f4"), ("e", "<i4")])
ORBIT_2X32_DTYPE = np.dtype([("x_head", "<f4"), ("x_tail", "<f4"), ("ex", "<i4"), ("ey", "<i4"),
                             ("y_head", "<f4"), ("y_tail", "<f4")])
assert ORBIT_HDR64_DTYPE.itemsize == 32 and REAL_HDR64.itemsize == 16
ORBIT_2X32_RC_DTYPE = np.dtype([("index", "<u8"), ("x_head", "<f4"), ("x_tail", "<f4"), ("ex", "<i4"), ("ey", "<i4"),
                                ("y_head", "<f4"), ("y_tail", "<f4")])
assert REAL_2X32.itemsize == 12 and ORBIT_2X32_DTYPE.itemsize == 24 and ORBIT_2X32_RC_DTYPE.itemsize == 32
assert ORBIT_HDR32_DTYPE.itemsize == 16 and LA_HDR32_DTYPE.itemsize == 68 and BLA_HDR32_DTYPE.itemsize == 44


def builtin_views():
    """Built-in view presets used by the BASELINE configs (data/views.json, see tools/extract_views.py)."""
    global _VIEWS
    if _VIEWS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "views.json")
        with open(path) as f:
            _VIEWS = {int(k): v for k, v in json.load(f).items()}
    return _VIEWS


class View:
    """Fractal::View(n) for a width x height window (aspect-squared bounding box + working precision)."""

    def __init__(self, min_x, min_y, max_x, max_y, width, height, num_iterations=8192, antialiasing=1):
        self._lib = _capi.inputs_lib()
        self.width, self.height = int(width), int(height)
        self.num_iterations = int(num_iterations)
        self.antialiasing = int(antialiasing)
        self._h = self._lib.fsh_view_create(min_x.encode(), min_y.encode(), max_x.encode(), max_y.encode(),
                                            self.width, self.height)
        if not self._h:
            raise RuntimeError("fsh_view_create failed")

    @classmethod
    def builtin(cls, n, width, height, antialiasing=None):
        v = builtin_views()[n]
        aa = v["gpuAntialiasing"] if antialiasing is None else antialiasing
        return cls(v["minX"], v["minY"], v["maxX"], v["maxY"], width, height, v["numIterations"], aa)

    @classmethod
    def load_im(cls, path, width, height, antialiasing=1):
        """A view from an Imagina ".im" location file (Fractal::LoadRefOrbit's location half: RefOrbitCalc.cpp:3425-3520,
        then RecenterViewCalc on {orbitX, orbitY, 2 / halfH}).  `.im_has_orbit` says whether the file also carried a
        reference orbit (not read: the orbit is recomputed), `.im_exp_bytes` which `long` wrote it."""
        lib = _capi.inputs_lib()
        limit, has_orbit, exp_bytes = C.c_uint64(0), C.c_int(0), C.c_int(0)
        h = lib.fsh_view_load_im(os.fsencode(path), int(width), int(height), C.byref(limit), C.byref(has_orbit),
                                 C.byref(exp_bytes))
        if not h:
            raise ValueError("%s: not an Imagina location file" % (path,))
        self = cls.__new__(cls)
        self._lib, self._h = lib, h
        self.width, self.height = int(width), int(height)
        self.num_iterations = int(limit.value)
        self.antialiasing = int(antialiasing)
        self.im_has_orbit, self.im_exp_bytes = bool(has_orbit.value), exp_bytes.value
        return self

    def save_im(self, path, exp_bytes=4):
        """The view as an Imagina location file, RefOrbitCalc::SaveOrbitResults(filename) (RefOrbitCalc.cpp:3117-3166).
        exp_bytes=4 writes what the reference's Windows build and Imagina write, 8 the reference's Linux build."""
        if self._lib.fsh_view_save_im(self._h, self.num_iterations, os.fsencode(path), int(exp_bytes)) != 0:
            raise OSError("could not write %s" % (path,))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_view_destroy(self._h)
            self._h = None

    @property
    def precision_bits(self):
        return self._lib.fsh_view_precision_bits(self._h)

    def bbox(self):
        out = []
        for k in range(4):
            buf = C.create_string_buffer(1 << 16)
            self._lib.fsh_view_bbox_str(self._h, k, buf, len(buf))
            out.append(buf.value.decode())
        return out

    def coords_direct_f64(self, aa=None):
        """{dx, dy, minX, maxY} as float64[4] (Cpu64 / direct kernels)."""
        aa = self.antialiasing if aa is None else aa
        out = np.zeros(4, np.float64)
        self._lib.fsh_view_coords_direct_f64(self._h, self.width * aa, self.height * aa, out.ctypes.data)
        return out

    def coords_direct_lp(self, kind, aa=None):
        """{cx = minX, cy = minY, dx, dy} for the Gpu1x32 ("1x32": float32[4]), Gpu2x32 ("2x32": float32[8] head/tail
        pairs), Gpu2x64 ("2x64": float64[8]), Gpu4x32 ("4x32": float32[16], x..w quadruples) and Gpu4x64 ("4x64":
        float64[16]) direct kernels."""
        aa = self.antialiasing if aa is None else aa
        k = {"1x32": 0, "2x32": 1, "2x64": 2, "4x32": 3, "4x64": 4}[kind]
        out = np.zeros((4, 8, 8, 16, 16)[k], np.float64 if k in (2, 4) else np.float32)
        self._lib.fsh_view_coords_direct_lp(self._h, self.width * aa, self.height * aa, k, out.ctypes.data)
        return out

    def coords_perturb_hdr32(self, orbit, aa=None):
        """{dx, dy, centerX, centerY} as 4 x {float mantissa, int32 exp} (8 x 4 bytes)."""
        aa = self.antialiasing if aa is None else aa
        out = np.zeros(4, REAL_HDR32)
        self._lib.fsh_view_coords_perturb_hdr32(self._h, orbit._h, self.width * aa, self.height * aa, out.ctypes.data)
        return out

    def coords_perturb(self, orbit, aa=None):
        """{dx, dy, centerX, centerY} in the orbit's type (hdr32 or hdr64 records)."""
        if not orbit.is64:
            return self.coords_perturb_hdr32(orbit, aa)
        aa = self.antialiasing if aa is None else aa
        out = np.zeros(4, REAL_HDR64)
        self._lib.fsh_view_coords_perturb_hdr64(self._h, orbit._h, self.width * aa, self.height * aa, out.ctypes.data)
        return out

    def coords_perturb_2x32(self, orbit, aa=None):
        """{dx, dy, centerX, centerY} as HDRFloat<CudaDblflt> records (head, tail, exp), mantissas in [0.5,1)."""
        aa = self.antialiasing if aa is None else aa
        out = np.zeros(4, REAL_2X32)
        o = orbit.source if isinstance(orbit, Orbit2x32) else orbit
        self._lib.fsh_view_coords_perturb_2x32(self._h, o._h, self.width * aa, self.height * aa, out.ctypes.data)
        return out

    def coords_direct_hdr(self, is64, aa=None):
        """{dx, dy, minX, maxY} as un-reduced HDRFloat records (CpuHDR32 / CpuHDR64)."""
        aa = self.antialiasing if aa is None else aa
        out = np.zeros(4, REAL_HDR64 if is64 else REAL_HDR32)
        fn = self._lib.fsh_view_coords_direct_hdr64 if is64 else self._lib.fsh_view_coords_direct_hdr32
        fn(self._h, self.width * aa, self.height * aa, out.ctypes.data)
        return out


class Orbit:
    """Reference orbit at the view centre (PerturbationResults<uint32_t, HDRFloat<float>, Disable> layout)."""

    def __init__(self, view, max_iter=None, periodicity=True, is64=False, compression_exp=None):
        """compression_exp: None = PerturbExtras::Disable; an int (reference default 20) = SimpleCompression."""
        self._lib = _capi.inputs_lib()
        self.view = view
        self.is64 = bool(is64)
        self.compressed = compression_exp is not None
        n = view.num_iterations if max_iter is None else max_iter
        self._h = self._lib.fsh_orbit_create_ex(view._h, 1 if is64 else 0, n, 1 if periodicity else 0,
                                                -1 if compression_exp is None else int(compression_exp))
        if not self._h:
            raise RuntimeError("fsh_orbit_create failed")
        self.count = self._lib.fsh_orbit_count(self._h)
        self.period = self._lib.fsh_orbit_period(self._h)

    @classmethod
    def load_im(cls, path, view):
        """The reference orbit stored in an Imagina ".im" file (RefOrbitCalc::LoadOrbitConst, RefOrbitCalc.cpp:3318-3423:
        LoadOrbitBin + DecompressMax): rebuilt from the file's waypoints, no high-precision iteration.  `view` is the
        view the orbit belongs to -- normally View.load_im(path, width, height).  HDRFloat<double> orbits come from
        files with Imagina's magic, HDRFloat<float> ones from "Sharks:)" files."""
        lib = _capi.inputs_lib()
        limit = C.c_uint64(0)
        h = lib.fsh_orbit_load_im(os.fsencode(path), C.byref(limit))
        if not h:
            raise ValueError("%s: no reference orbit this reader takes" % (path,))
        self = cls.__new__(cls)
        self._lib, self._h, self.view = lib, h, view
        self.is64 = bool(lib.fsh_orbit_is64(h))
        self.compressed = False
        self.count = lib.fsh_orbit_count(h)
        self.period = lib.fsh_orbit_period(h)
        self.im_iteration_limit = int(limit.value)
        return self

    def save_im(self, path, compression_exp=20, exp_bytes=4):
        """The view's location and this orbit under "max compression" as an Imagina ".im" file
        (RefOrbitCalc::SaveOrbitResults(results, filename), RefOrbitCalc.cpp:3039-3115).  compression_exp: the
        reference's Fractal::CompressionError::Low default."""
        if self._lib.fsh_orbit_save_im(self._h, self.view.num_iterations, int(compression_exp), os.fsencode(path),
                                       int(exp_bytes)) != 0:
            raise OSError("could not write %s" % (path,))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_orbit_destroy(self._h)
            self._h = None

    def scale_entries(self, indices, exp2):
        """Test hook (fsh_orbit_scale_entries): entries `indices` scaled by 2^exp2 -- an orbit-shaped input with period
        boundaries where a test wants them; not the orbit of any view any more."""
        idx = np.ascontiguousarray(indices, np.uint64)
        ex = np.ascontiguousarray(exp2, np.int32)
        assert idx.shape == ex.shape
        return int(self._lib.fsh_orbit_scale_entries(self._h, idx.ctypes.data, ex.ctypes.data, idx.size))

    @property
    def data_ptr(self):
        return self._lib.fsh_orbit_data_hdr64(self._h) if self.is64 else self._lib.fsh_orbit_data_hdr32(self._h)

    @property
    def compressed_count(self):
        return self._lib.fsh_orbit_compressed_count(self._h)

    @property
    def compressed_data_ptr(self):
        if self.is64:
            return self._lib.fsh_orbit_compressed_data_hdr64(self._h)
        return self._lib.fsh_orbit_compressed_data_hdr32(self._h)

    def orbit_low(self):
        out = np.zeros(2, REAL_HDR64 if self.is64 else REAL_HDR32)
        (self._lib.fsh_orbit_low_hdr64 if self.is64 else self._lib.fsh_orbit_low_hdr32)(self._h, out.ctypes.data)
        return out

    def max_radius(self):
        """PerturbationResults::GetMaxRadius() as an ABI record (the blaSize argument of BLAS::Init)."""
        out = np.zeros(1, REAL_HDR64 if self.is64 else REAL_HDR32)
        (self._lib.fsh_orbit_max_radius_hdr64 if self.is64 else self._lib.fsh_orbit_max_radius_hdr32)(
            self._h, out.ctypes.data)
        return out

    # PerturbExtras::Bad form (scaled kernels): the HDRFloat<float> orbit with its flags + its binary32 copy
    @property
    def bad_data_ptr(self):
        return self._lib.fsh_orbit_data_hdr32_bad(self._h)

    @property
    def bad_f32_data_ptr(self):
        return self._lib.fsh_orbit_data_f32_bad(self._h)

    @property
    def bad_count(self):
        return self._lib.fsh_orbit_bad_count(self._h)

    def entries(self):
        dt = ORBIT_HDR64_DTYPE if self.is64 else ORBIT_HDR32_DTYPE
        buf = (C.c_uint8 * (self.count * dt.itemsize)).from_address(self.data_ptr)
        return np.frombuffer(buf, dtype=dt).copy()


class Orbit2x32:
    """The HDRFloat<CudaDblflt> twin of an HDRFloat<double> orbit, converted entry by entry the way
    PerturbationResults::CopyPerturbationResults does (PerturbationResults.cpp:239-347).  From a SimpleCompression
    source the waypoints are what is converted (:265-268); the full orbit then only exists on the GPU, where it is
    rebuilt in 2x32 arithmetic."""

    is64 = False

    def __init__(self, orbit64):
        if not orbit64.is64:
            raise ValueError("the 2x32 orbit is derived from the HDRFloat<double> orbit")
        lib = self._lib = _capi.inputs_lib()
        self.source = orbit64
        self.view = orbit64.view
        self.count, self.period = orbit64.count, orbit64.period
        self.compressed = orbit64.compressed
        if self.compressed:
            self.compressed_count = orbit64.compressed_count
            self._rc = np.zeros(self.compressed_count, ORBIT_2X32_RC_DTYPE)
            lib.fsh_convert_orbit_rc_hdr64_to_2x32(orbit64.compressed_data_ptr, self.compressed_count,
                                                   self._rc.ctypes.data)
            self._data = None
        else:
            self._data = np.zeros(self.count, ORBIT_2X32_DTYPE)
            lib.fsh_convert_orbit_hdr64_to_2x32(orbit64.data_ptr, self.count, self._data.ctypes.data)

    @property
    def data_ptr(self):
        if self._data is None:
            raise ValueError("a SimpleCompression 2x32 orbit has no host-side uncompressed form")
        return self._data.ctypes.data

    @property
    def compressed_data_ptr(self):
        return self._rc.ctypes.data

    def waypoints(self):
        return self._rc.copy()

    def orbit_low(self):
        out = np.zeros(2, REAL_2X32)
        self._lib.fsh_orbit_low_2x32(self.source._h, out.ctypes.data)
        return out

    def entries(self):
        if self._data is None:
            raise ValueError("a SimpleCompression 2x32 orbit has no host-side uncompressed form")
        return self._data.copy()


class LATable2x32:
    """The HDRFloat<CudaDblflt> twin of an HDRFloat<double> LAv2 table (LAReference::CopyLAReference,
    LAReference.h:163-213): 104-byte records, the stages unchanged, ATInfo converted field by field."""

    def __init__(self, la64):
        if not la64.is64:
            raise ValueError("the 2x32 LA table is derived from the HDRFloat<double> table")
        if not la64.use_small_exponents:
            raise ValueError("build the source table with use_small_exponents=True (RefOrbitCalc.cpp:2346)")
        lib = _capi.inputs_lib()
        self.source = la64
        self.count, self.stage_count = la64.count, la64.stage_count
        self.is_valid, self.use_at = la64.is_valid, la64.use_at
        self._las = np.zeros((max(self.count, 1), 104), np.uint8)
        lib.fsh_convert_la_hdr64_to_2x32(la64.las_ptr, self.count, self._las.ctypes.data)
        self._stages = la64.stages()
        self.at = _capi.At2x32()
        lib.fsh_convert_at_hdr64_to_2x32(C.addressof(la64.at), C.addressof(self.at))

    @property
    def las_ptr(self):
        return self._las.ctypes.data

    @property
    def stages_ptr(self):
        return self._stages.ctypes.data

    def records(self):
        return self._las[: self.count].copy()

    def stages(self):
        return self._stages.copy()


class LATableU64:
    """The IterType = uint64_t form of an LAv2 table (LAInfoDeep<uint64_t,..>, LAStageInfo<uint64_t>, ATInfo<uint64_t,..>):
    what a FractalShark build with 64-bit iteration counts hands to InitializePerturb<uint64_t,...>.  Built by widening
    the counts of a uint32_t table; the coefficient fields are byte-identical (include/fs_layout.h)."""

    def __init__(self, la):
        self.source = la
        self.count, self.stage_count = la.count, la.stage_count
        self.is_valid, self.use_at = la.is_valid, la.use_at
        rec32 = np.ascontiguousarray(la.records()).view(np.uint8).reshape(la.count, -1)
        w32 = rec32.shape[1]                      # 68 / 128 / 104
        coeff = w32 - 8                           # bytes before StepLength
        w64 = (coeff + 7) // 8 * 8 + 16           # 80 / 136 / 112
        out = np.zeros((max(la.count, 1), w64), np.uint8)
        out[: la.count, :coeff] = rec32[:, :coeff]
        steps = rec32[:, coeff:].copy().view(np.uint32).astype(np.uint64)
        out[: la.count, w64 - 16:] = steps.view(np.uint8).reshape(la.count, 16)
        self._las = out
        self._stages = la.stages().astype(np.uint64)
        at32 = bytes(la.at)
        step_bytes = 8 if len(at32) == 232 else 4  # the double record pads StepLength to 8 already
        at64 = np.zeros((len(at32) - step_bytes + 8 + 7) // 8 * 8, np.uint8)
        at64[:8] = np.frombuffer(np.uint64(la.at.StepLength).tobytes(), np.uint8)
        at64[8:8 + len(at32) - step_bytes] = np.frombuffer(at32[step_bytes:], np.uint8)
        self._at = at64
        self.at = (C.c_uint8 * len(at64)).from_buffer(self._at)

    @property
    def las_ptr(self):
        return self._las.ctypes.data

    @property
    def stages_ptr(self):
        return self._stages.ctypes.data


class LATable:
    """LAv2 table (LAReference<uint32_t, HDRFloat<float>, float, Disable>)."""

    def __init__(self, orbit, host_threads=8, use_small_exponents=False):
        """use_small_exponents: the reference's UsingDblflt flag -- set it for a table that will be converted to
        2x32 (LATable2x32); it caps the AT escape radius at 2^32 (LAInfoDeep.h:484-496)."""
        self._lib = _capi.inputs_lib()
        self.orbit = orbit
        self.is64 = orbit.is64
        self.use_small_exponents = bool(use_small_exponents)
        self._h = self._lib.fsh_la_create_ex(orbit._h, host_threads, 1 if use_small_exponents else 0)
        if not self._h:
            raise RuntimeError("fsh_la_create failed")
        self.count = self._lib.fsh_la_count(self._h)
        self.stage_count = self._lib.fsh_la_stage_count(self._h)
        self.is_valid = bool(self._lib.fsh_la_is_valid(self._h))
        self.use_at = bool(self._lib.fsh_la_use_at(self._h))
        self.at = _capi.AtHdr64() if self.is64 else _capi.AtHdr32()
        self._lib.fsh_la_at(self._h, C.addressof(self.at))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_la_destroy(self._h)
            self._h = None

    @property
    def las_ptr(self):
        return self._lib.fsh_la_data(self._h)

    @property
    def stages_ptr(self):
        return self._lib.fsh_la_stages(self._h)

    def records(self):
        if self.is64:
            buf = (C.c_uint8 * (self.count * 128)).from_address(self.las_ptr)
            return np.frombuffer(buf, dtype=np.uint8).reshape(-1, 128).copy()
        buf = (C.c_uint8 * (self.count * 68)).from_address(self.las_ptr)
        return np.frombuffer(buf, dtype=LA_HDR32_DTYPE).copy()

    def stages(self):
        buf = (C.c_uint32 * (self.stage_count * 2)).from_address(self.stages_ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(-1, 2).copy()


class BLATable:
    """BLA table (BLAS<uint32_t, HDRFloat<float>>)."""

    def __init__(self, orbit):
        self._lib = _capi.inputs_lib()
        self.orbit = orbit
        self.is64 = orbit.is64
        self._h = self._lib.fsh_bla_create(orbit._h)
        if not self._h:
            raise RuntimeError("fsh_bla_create failed")
        self.num_levels = self._lib.fsh_bla_num_levels(self._h)
        self.lm2 = self._lib.fsh_bla_lm2(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_bla_destroy(self._h)
            self._h = None

    @property
    def level_ptrs(self):
        return self._lib.fsh_bla_level_ptrs(self._h)

    @property
    def level_sizes(self):
        return self._lib.fsh_bla_level_sizes(self._h)

    def sizes(self):
        buf = (C.c_uint64 * self.num_levels).from_address(self.level_sizes)
        return list(buf)

    def level(self, l):
        """Records of level l as raw bytes (n, 44) / (n, 88)."""
        n = self.sizes()[l]
        rec = 88 if self.is64 else 44
        if n == 0:
            return np.zeros((0, rec), np.uint8)
        ptrs = (C.c_void_p * self.num_levels).from_address(self.level_ptrs)
        buf = (C.c_uint8 * (n * rec)).from_address(ptrs[l])
        return np.frombuffer(buf, dtype=np.uint8).reshape(n, rec).copy()


class OrbitF64:
    """Plain-double reference orbit + its BLA table (PerturbationResults<uint32_t,double,Disable>, BLAS<uint32_t,double>):
    the inputs of Cpu64PerturbedBLA / Gpu1x64PerturbedBLA."""

    def __init__(self, view, max_iter=None, periodicity=True):
        self._lib = _capi.inputs_lib()
        self.view = view
        n = view.num_iterations if max_iter is None else max_iter
        self._h = self._lib.fsh_orbit_f64_create(view._h, n, 1 if periodicity else 0)
        self.count = self._lib.fsh_orbit_f64_count(self._h)
        self.period = self._lib.fsh_orbit_f64_period(self._h)
        self.num_levels = self._lib.fsh_orbit_f64_bla_num_levels(self._h)
        self.lm2 = self._lib.fsh_orbit_f64_bla_lm2(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_orbit_f64_destroy(self._h)
            self._h = None

    @property
    def data_ptr(self):
        return self._lib.fsh_orbit_f64_data(self._h)

    @property
    def bad_data_ptr(self):
        return self._lib.fsh_orbit_f64_data_bad(self._h)

    @property
    def bad_f32_data_ptr(self):
        return self._lib.fsh_orbit_f64_data_f32_bad(self._h)

    @property
    def level_ptrs(self):
        return self._lib.fsh_orbit_f64_bla_level_ptrs(self._h)

    @property
    def level_sizes(self):
        return self._lib.fsh_orbit_f64_bla_level_sizes(self._h)

    def coords(self, aa=None):
        aa = self.view.antialiasing if aa is None else aa
        out = np.zeros(4, np.float64)
        self._lib.fsh_view_coords_perturb_f64(self.view._h, self._h, self.view.width * aa, self.view.height * aa,
                                              out.ctypes.data)
        return out


# ---- plain (non-HDR) LAv2 inputs: Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*
def _plain_dtypes(kind):
    f = {"f32": "<f4", "f64": "<f8"}.get(kind)
    if kind == "2x32":
        real = [("head", "<f4"), ("tail", "<f4")]
        cplx = np.dtype([("re_head", "<f4"), ("re_tail", "<f4"), ("im_head", "<f4"), ("im_tail", "<f4")])
        orbit = np.dtype([("x_head", "<f4"), ("x_tail", "<f4"), ("y_head", "<f4"), ("y_tail", "<f4")])
        real = np.dtype(real)
    else:
        real = np.dtype(f)
        cplx = np.dtype([("re", f), ("im", f)])
        orbit = np.dtype([("x", f), ("y", f)])
    la = np.dtype([("Ref", cplx), ("ZCoeff", cplx), ("CCoeff", cplx), ("LAThreshold", real), ("LAThresholdC", real),
                   ("MinMag", real), ("StepLength", "<u4"), ("NextStageLAIndex", "<u4")])
    at = np.dtype([("StepLength", "<u4"), ("ThresholdC", real), ("SqrEscapeRadius", real), ("RefC", cplx),
                   ("ZCoeff", cplx), ("CCoeff", cplx), ("InvZCoeff", cplx), ("CCoeffSqrInvZCoeff", cplx),
                   ("CCoeffInvZCoeff", cplx), ("CCoeffNormSqr", real), ("RefCNormSqr", real), ("factor", real)],
                  align=(kind == "f64"))
    return orbit, la, at, real


class PlainInputs:
    """Orbit + LAv2 table + coordinates for the non-HDR LAv2 algorithms.

    kind "f32": PerturbationResults<u32,float,Disable> + LAReference<u32,float,float,Disable> built in binary32
    (Gpu1x32PerturbedLAv2*); "f64": the same in binary64 (Gpu1x64PerturbedLAv2*); "2x32": the binary64 inputs converted
    field by field to CudaDblflt<MattDblflt> (Gpu2x32PerturbedLAv2*, Fractal.cpp:2771-2772)."""

    def __init__(self, view, kind, host_threads=1, periodicity=True, max_iter=None, compression_exp=None):
        """compression_exp: None = PerturbExtras::Disable; an int (reference default 20) = SimpleCompression (the
        Gpu*PerturbedRCLAv2* algorithms): `waypoints()` is then what is uploaded, `orbit()` what the host's
        RuntimeDecompressor makes of it (for "2x32": of the binary64 source; the GPU rebuilds its own in 2x32)."""
        assert kind in ("f32", "f64", "2x32")
        lib = self._lib = _capi.inputs_lib()
        self.view, self.kind = view, kind
        self.compressed = compression_exp is not None
        self._h = lib.fsh_plain_create_ex(view._h, 0 if kind == "f32" else 1,
                                          view.num_iterations if max_iter is None else max_iter,
                                          1 if periodicity else 0, host_threads,
                                          -1 if compression_exp is None else int(compression_exp))
        if not self._h:
            raise RuntimeError("fsh_plain_create failed")
        self._adopt()

    @classmethod
    def load_im(cls, path, view, kind=None, host_threads=1):
        """The reference orbit of an Imagina ".im" file written for a non-ExtendedRange type (ReferenceHeader::ExtendedRange
        false; float for the "Sharks:)" magic, double for Imagina's -- RefOrbitCalc.cpp:3386-3412), rebuilt from the file's
        waypoints (LoadOrbitBin + DecompressMax), with the LAv2 table built from it.  kind: None = the file's type
        ("f32" / "f64"); "2x32" converts a double file's inputs to CudaDblflt like the constructor does."""
        lib = _capi.inputs_lib()
        limit = C.c_uint64(0)
        h = lib.fsh_plain_load_im(os.fsencode(path), C.byref(limit), int(host_threads))
        if not h:
            raise ValueError("%s: no plain-type reference orbit this reader takes" % (path,))
        file_kind = "f32" if lib.fsh_plain_kind(h) == 0 else "f64"
        if kind is None:
            kind = file_kind
        if (kind == "f32") != (file_kind == "f32"):
            lib.fsh_plain_destroy(h)
            raise ValueError("%s holds a %s orbit" % (path, file_kind))
        self = cls.__new__(cls)
        self._lib, self._h, self.view, self.kind, self.compressed = lib, h, view, kind, False
        self.im_iteration_limit = int(limit.value)
        self._adopt()
        return self

    def save_im(self, path, compression_exp=20, exp_bytes=4):
        """The view's location and this orbit under "max compression" as an Imagina ".im" file, the form
        RefOrbitCalc::SaveOrbitResults(results, filename) writes for PerturbationResults<IterType, float | double, ...>
        (RefOrbitCalc.cpp:3039-3115: ExtendedRange = false, waypoints as two doubles and the index field)."""
        if self._lib.fsh_plain_save_im(self._h, self.view.num_iterations, int(compression_exp), os.fsencode(path),
                                       int(exp_bytes)) != 0:
            raise OSError("could not write %s" % (path,))

    def _adopt(self):
        lib, view, kind = self._lib, self.view, self.kind
        src_kind = "f32" if kind == "f32" else "f64"
        o_dt, la_dt, at_dt, real_dt = _plain_dtypes(src_kind)
        self.count = int(lib.fsh_plain_orbit_count(self._h))
        self.period = int(lib.fsh_plain_orbit_period(self._h))
        self.la_count = int(lib.fsh_plain_la_count(self._h))
        self.stage_count = int(lib.fsh_plain_la_stage_count(self._h))
        self.is_valid = bool(lib.fsh_plain_la_is_valid(self._h))
        self.use_at = bool(lib.fsh_plain_la_use_at(self._h))

        def grab(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_uint8 * (n * dt.itemsize)).from_address(ptr), dtype=dt).copy()

        orbit = grab(lib.fsh_plain_orbit_data(self._h), self.count, o_dt)
        las = grab(lib.fsh_plain_la_data(self._h), self.la_count, la_dt)
        self._stages = grab(lib.fsh_plain_la_stages(self._h), self.stage_count, np.dtype([("LAIndex", "<u4"),
                                                                                          ("MacroItCount", "<u4")]))
        at = np.zeros(1, at_dt)
        lib.fsh_plain_la_at(self._h, at.ctypes.data)
        aa = view.antialiasing
        coords = np.zeros(4, real_dt)
        lib.fsh_plain_coords(view._h, self._h, view.width * aa, view.height * aa, coords.ctypes.data)
        if self.compressed:
            self.compressed_count = int(lib.fsh_plain_compressed_count(self._h))
            rc_dt = np.dtype([("index", "<u8")] + [(n, o_dt.fields[n][0]) for n in o_dt.names])
            rc = grab(lib.fsh_plain_compressed_data(self._h), self.compressed_count, rc_dt)
            low = np.zeros(4, real_dt)  # {OrbitXLow, OrbitYLow, -, -}
            lib.fsh_plain_orbit_low(self._h, low.ctypes.data)
            if kind == "2x32":
                o2 = _plain_dtypes("2x32")[0]
                self._rc = np.zeros(self.compressed_count, np.dtype([("index", "<u8")] + [(n, "<f4") for n in o2.names]))
                lib.fsh_convert_orbit_rc_f64_to_p2x32(rc.ctypes.data, self.compressed_count, self._rc.ctypes.data)
                self._low = np.zeros(4, _plain_dtypes("2x32")[3])
                lib.fsh_convert_coords_f64_to_p2x32(low.ctypes.data, self._low.ctypes.data)
            else:
                self._rc, self._low = rc, low
        if kind == "2x32":
            o2, la2, at2, real2 = _plain_dtypes("2x32")
            self._orbit = np.zeros(self.count, o2)
            lib.fsh_convert_orbit_f64_to_p2x32(orbit.ctypes.data, self.count, self._orbit.ctypes.data)
            self._las = np.zeros(max(self.la_count, 1), la2)
            lib.fsh_convert_la_f64_to_p2x32(las.ctypes.data, self.la_count, self._las.ctypes.data)
            self._at = np.zeros(1, at2)
            lib.fsh_convert_at_f64_to_p2x32(at.ctypes.data, self._at.ctypes.data)
            self._coords = np.zeros(4, real2)
            lib.fsh_convert_coords_f64_to_p2x32(coords.ctypes.data, self._coords.ctypes.data)
        else:
            self._orbit, self._las, self._at, self._coords = orbit, (las if self.la_count else np.zeros(1, la_dt)), at, coords

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsh_plain_destroy(self._h)
            self._h = None

    orbit_ptr = property(lambda self: self._orbit.ctypes.data)
    las_ptr = property(lambda self: self._las.ctypes.data)
    stages_ptr = property(lambda self: self._stages.ctypes.data if self.stage_count else None)
    at_ptr = property(lambda self: self._at.ctypes.data)
    coords_ptr = property(lambda self: self._coords.ctypes.data)
    compressed_data_ptr = property(lambda self: self._rc.ctypes.data)

    def waypoints(self):
        return self._rc.copy()

    def orbit_low(self):
        """{OrbitXLow, OrbitYLow} in the type of `kind`."""
        return self._low[:2].copy()

    def orbit(self):
        return self._orbit.copy()

    def las(self):
        return self._las[: self.la_count].copy()

    def stages(self):
        return self._stages.copy()

    def at(self):
        return self._at.copy()

    def coords(self):
        return self._coords.copy()
