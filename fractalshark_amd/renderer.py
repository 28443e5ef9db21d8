"""Python mirror of the reference's `GPURenderer` (FractalSharkLib/GPU_Render.h:20-227) over the C ABI.

Method names, argument order and error behaviour follow the reference class so the tests read like calls
from Fractal.cpp (every method returns the uint32 error code, 0 = success; template parameters of the
reference become the `T=`/`Mode=` keyword tags).  All compute happens in libfsmi355.so (hand-written HIP);
this module owns no arithmetic and there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi

# numeric type tags (include/fsmi355.h)
T_F32, T_F64, T_2X32, T_HDR32, T_HDR64, T_HDR2X32, T_2X64, T_4X32, T_4X64 = range(9)
# LAv2Mode (RenderAlgorithm.h:12-17)
LAV2_FULL, LAV2_PO, LAV2_LAO = range(3)
# parity (include/fsmi355.h)
PARITY_CPU, PARITY_CPU_GPUSTAGE = range(2)
FS_ERR_UNSUPPORTED = 10100
# A/B flags of fs_set_kernel_variant (include/fsmi355.h)
VARIANT_LDS_ORBIT, VARIANT_REFILL, VARIANT_WIDE_COUNTERS, VARIANT_NATURAL_TILE_ORDER = 0x100, 0x200, 0x400, 0x800

NB_THREADS_W = 16  # GPU_Render.h:116-120: part of the contract (ItersMemoryContainer pads with them)
NB_THREADS_H = 8


class GPURenderer:
    def __init__(self, device=0):
        self._lib = _capi.render_lib()
        self._h = self._lib.fs_create(int(device))
        if not self._h:
            raise MemoryError("fs_create failed")
        self._cbs = []

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):  # (a member of a GPURendererGroup belongs to the group)
                self._lib.fs_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # ---- static members
    @staticmethod
    def device_count():
        """HIP devices visible to this process (fs_device_count)."""
        return int(_capi.render_lib().fs_device_count())

    @staticmethod
    def TestCudaIsWorking():
        """Non-zero = working (GPU_Render.cu:100-123)."""
        return _capi.render_lib().fs_test_device_is_working()

    @staticmethod
    def ConvertErrorToString(err):
        return _capi.render_lib().fs_error_string(int(err)).decode()

    # ---- memory
    def InitializeMemory(self, w, h, antialiasing, palInterleaved, palIters, paletteAuxDepth, paletteGeneration,
                         expectedReuse, iter_bytes=4):
        """w, h already include antialiasing.  palInterleaved: uint16[palIters,4] RGBA or None."""
        if palInterleaved is not None:
            pal = np.ascontiguousarray(palInterleaved, dtype=np.uint16)
            self._pal_keepalive = pal
            pal_ptr = pal.ctypes.data
        else:
            pal_ptr = None
        self._iter_bytes = int(iter_bytes)
        return self._lib.fs_init_memory(self._h, w, h, antialiasing, iter_bytes, pal_ptr, palIters, paletteAuxDepth,
                                        paletteGeneration, 1 if expectedReuse else 0)

    def ClearMemory(self):
        return self._lib.fs_clear(self._h)

    def SetRowBands(self, band_first_row, band_rows, band_stride_rows):
        return self._lib.fs_set_row_bands(self._h, band_first_row, band_rows, band_stride_rows)

    @property
    def compute_stream(self):
        """The renderer's compute stream as a raw hipStream_t (int), e.g. for torch.cuda.ExternalStream."""
        return self._lib.fs_compute_stream(self._h) or 0

    def SetExternalIterBuffer(self, device_ptr, capacity_bytes=0):
        return self._lib.fs_set_external_iter_buffer(self._h, device_ptr, int(capacity_bytes))

    def CopyBandsToHost(self, host_frame_ptr, device_iters=None, stream=None):
        """fs_copy_bands_to_host: this renderer's row bands -> their rows of a whole-frame host buffer (raw pointer), over
        this device's own PCIe link, asynchronously on `stream` (raw hipStream_t; None = the compute stream)."""
        return self._lib.fs_copy_bands_to_host(self._h, device_iters, host_frame_ptr, stream)

    def GetWidth(self):
        return self._lib.fs_get_width(self._h)

    def GetHeight(self):
        return self._lib.fs_get_height(self._h)

    @property
    def rounded_width(self):
        return self._lib.fs_rounded_width(self._h)

    @property
    def local_rows(self):
        return self._lib.fs_local_rows(self._h)

    @property
    def device_iter_buffer(self):
        return self._lib.fs_device_iter_buffer(self._h)

    # ---- uploads
    def InitializePerturb(self, GenerationNumber1, Perturb1, GenerationNumber2=0, Perturb2=None,
                          LaReferenceHost=None, T=None, iter_bytes=4):
        """Perturb1: inputs.Orbit (or anything with data_ptr/count/period); LaReferenceHost: inputs.LATable.
        T defaults to the orbit's type (T_HDR32 / T_HDR64 / T_HDR2X32).  iter_bytes = sizeof(IterType): with 8,
        LaReferenceHost must be an inputs.LATableU64."""
        if T is None:
            if type(Perturb1).__name__ == "Orbit2x32":
                T = T_HDR2X32
            else:
                T = T_HDR64 if getattr(Perturb1, "is64", False) else T_HDR32
        if getattr(Perturb1, "compressed", False):
            # PerturbExtras::SimpleCompression: hand over the waypoints, like the reference's *RC* algorithms
            low = Perturb1.orbit_low()
            err = self._lib.fs_upload_orbit_compressed(self._h, GenerationNumber1, T, iter_bytes,
                                                       Perturb1.compressed_data_ptr,
                                                       Perturb1.compressed_count, Perturb1.count, Perturb1.period,
                                                       low[0:1].ctypes.data, low[1:2].ctypes.data)
        else:
            err = self._lib.fs_upload_orbit(self._h, GenerationNumber1, T, iter_bytes, Perturb1.data_ptr, Perturb1.count,
                                            Perturb1.count, Perturb1.period)
        if err:
            return err
        if LaReferenceHost is not None:
            la = LaReferenceHost
            if (iter_bytes == 8) != (type(la).__name__ == "LATableU64"):
                raise ValueError("iter_bytes=8 needs an inputs.LATableU64 (and only then)")
            err = self._lib.fs_upload_la(self._h, GenerationNumber1, T, iter_bytes, la.las_ptr, la.count, la.stages_ptr,
                                         la.stage_count, 1 if la.is_valid else 0, 1 if la.use_at else 0,
                                         C.addressof(la.at))
        return err

    def InitializePerturbPlain(self, GenerationNumber1, plain, with_la=True):
        """InitializePerturb<IterType, T, T, Disable, T> for a non-HDR T: plain = inputs.PlainInputs (kind f32 -> float,
        f64 -> double, 2x32 -> CudaDblflt<MattDblflt>), i.e. the inputs of Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*."""
        T = {"f32": T_F32, "f64": T_F64, "2x32": T_2X32}[plain.kind]
        if getattr(plain, "compressed", False):  # PerturbExtras::SimpleCompression: Gpu*PerturbedRCLAv2*
            low = plain.orbit_low()
            err = self._lib.fs_upload_orbit_compressed(self._h, GenerationNumber1, T, 4, plain.compressed_data_ptr,
                                                       plain.compressed_count, plain.count, plain.period,
                                                       low[0:1].ctypes.data, low[1:2].ctypes.data)
        else:
            err = self._lib.fs_upload_orbit(self._h, GenerationNumber1, T, 4, plain.orbit_ptr, plain.count, plain.count,
                                            plain.period)
        if err or not with_la:
            return err
        return self._lib.fs_upload_la(self._h, GenerationNumber1, T, 4, plain.las_ptr, plain.la_count,
                                      plain.stages_ptr, plain.stage_count, 1 if plain.is_valid else 0,
                                      1 if plain.use_at else 0, plain.at_ptr)

    def RenderPerturbLAv2Plain(self, plain, n_iterations, Mode=LAV2_FULL):
        """RenderPerturbLAv2<IterType, T, T, Mode, Disable> for the type of `plain` (see InitializePerturbPlain)."""
        T = {"f32": T_F32, "f64": T_F64, "2x32": T_2X32}[plain.kind]
        return self._lib.fs_render_lav2(self._h, T, Mode, PARITY_CPU, plain.coords_ptr, int(n_iterations))

    # ---- renders (asynchronous on the compute stream)
    @staticmethod
    def _pack_coords(T, vals):
        """vals: 4 x (mantissa, exp) pairs -> ABI records of the numeric type T."""
        if T == T_HDR64:
            dt = np.dtype([("m", "<f8"), ("e", "<i4"), ("pad_", "<i4")])
            return np.array([(float(m), int(e), 0) for m, e in vals], dtype=dt)
        if T == T_HDR2X32:  # (head, tail, exp) triples
            dt = np.dtype([("head", "<f4"), ("tail", "<f4"), ("e", "<i4")])
            return np.array([(float(h), float(t), int(e)) for h, t, e in vals], dtype=dt)
        dt = np.dtype([("m", "<f4"), ("e", "<i4")])
        return np.array([(float(m), int(e)) for m, e in vals], dtype=dt)

    def RenderPerturbLAv2(self, algorithm, cx, cy, dx, dy, centerX, centerY, n_iterations, T=T_HDR32,
                          Mode=LAV2_FULL, parity=PARITY_CPU):
        """cx, cy are unused by the kernels (as in the reference).  dx..centerY: (mantissa, exp) pairs."""
        co = self._pack_coords(T, [dx, dy, centerX, centerY])
        return self._lib.fs_render_lav2(self._h, T, Mode, parity, co.ctypes.data, int(n_iterations))

    def RenderPerturbBLA(self, algorithm, results, blas, cx, cy, dx, dy, centerX, centerY, n_iterations,
                         iteration_precision=1, T=None):
        """Uploads orbit and table on every call, like the reference (GPU_Render.cu:1464-1479)."""
        if T is None:
            T = T_HDR64 if getattr(results, "is64", False) else T_HDR32
        err = self._lib.fs_upload_orbit(self._h, 0, T, 4, results.data_ptr, results.count, results.count,
                                        results.period)
        if err:
            return err
        if blas is not None:
            err = self._lib.fs_upload_bla(self._h, T, blas.level_ptrs, blas.level_sizes, blas.num_levels, blas.lm2)
        else:
            err = self._lib.fs_upload_bla(self._h, T, None, None, 0, 0)
        if err:
            return err
        co = self._pack_coords(T, [dx, dy, centerX, centerY])
        return self._lib.fs_render_bla(self._h, T, co.ctypes.data, int(n_iterations))

    def RenderPerturbBLAScaled(self, algorithm, double_perturb, float_perturb, cx, cy, dx, dy, centerX, centerY,
                               n_iterations, iteration_precision=1, T=T_HDR32):
        """GpuHDRx32PerturbedScaled (T_HDR32: double_perturb = inputs.Orbit, coords (mantissa, exp) pairs) or
        Gpu1x32PerturbedScaled (T_F64: double_perturb = inputs.OrbitF64, coords doubles).  The PerturbExtras::Bad form of
        the orbit and its binary32 copy are taken from it; uploaded on every call like the reference
        (GPU_Render.cu:1324-1345)."""
        ob = double_perturb
        err = self._lib.fs_upload_orbit_scaled(self._h, T, 4, ob.bad_data_ptr, (float_perturb or ob).bad_f32_data_ptr,
                                               ob.count, ob.period)
        if err:
            return err
        if T == T_F64:
            co = np.array([dx, dy, centerX, centerY], dtype=np.float64)
        else:
            co = self._pack_coords(T, [dx, dy, centerX, centerY])
        return self._lib.fs_render_scaled(self._h, T, co.ctypes.data, int(n_iterations))

    def BuildBLAOnDevice(self, orbit, T=None):
        """BLAS::Init(count, maxRadius) on the device for the orbit last uploaded (fs_build_bla): the table stays in HBM."""
        if T is None:
            T = T_HDR64 if orbit.is64 else T_HDR32
        mr = orbit.max_radius()
        return self._lib.fs_build_bla(self._h, T, mr.ctypes.data)

    def BuildLAOnDevice(self, orbit, use_small_exponents=False, T=None, host_fallback=True, host_threads=1):
        """LAReference::GenerateApproximationData on the device for the orbit last uploaded (fs_build_la): all stages and
        the ATInfo stay in HBM, installed as the renderer's table.  An orbit of at most 64 steps in which no period is
        found gets the reference's two records and a table that is NOT valid (LAReference.cpp:135-140); the kernels ignore
        it, as they do in FractalShark.  The degenerate inputs the device builder leaves to the host (FS_ERR_UNSUPPORTED: an
        orbit of fewer than three entries, a first step with a zero ZCoeff -- microseconds of host work) are built by the
        host builder and uploaded with fs_upload_la, which is what FractalShark itself does for every table
        (host_fallback=False returns the error code instead).  host_threads > 1: stage 0 as CreateLAFromOrbitMT builds it on a
        host with that many hardware threads (fs_build_la_mt)."""
        if T is None:
            T = T_HDR64 if orbit.is64 else T_HDR32
        mr = orbit.max_radius()
        err = self._lib.fs_build_la_mt(self._h, T, mr.ctypes.data, 1 if use_small_exponents else 0, int(host_threads))
        if err == FS_ERR_UNSUPPORTED and host_fallback:
            from . import inputs
            la = inputs.LATable(orbit, host_threads=host_threads, use_small_exponents=use_small_exponents)
            err = self._lib.fs_upload_la(self._h, 0, T, 4, la.las_ptr, la.count, la.stages_ptr, la.stage_count,
                                         1 if la.is_valid else 0, 1 if la.use_at else 0, C.addressof(la.at))
        return err

    def read_la(self, is64=False):
        """Device-resident LA table -> (records uint8[n, 68|128], stages uint32[k, 2], at bytes, use_at, is_valid)."""
        n, k, ua, iv = C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_int(0)
        err = self._lib.fs_la_counts(self._h, C.byref(n), C.byref(k), C.byref(ua), C.byref(iv))
        if err:
            raise RuntimeError(self.ConvertErrorToString(err))
        rec = 128 if is64 else 68
        las = np.zeros((n.value, rec), np.uint8)
        stages = np.zeros((k.value, 2), np.uint32)
        at = _capi.AtHdr64() if is64 else _capi.AtHdr32()
        err = self._lib.fs_read_la(self._h, las.ctypes.data, n.value, stages.ctypes.data, k.value, C.addressof(at))
        if err:
            raise RuntimeError(self.ConvertErrorToString(err))
        return las, stages, bytes(at), bool(ua.value), bool(iv.value)

    def read_bla_levels(self, is64=False):
        """Device-resident BLA table -> list of (n, 44|88) uint8 arrays per level (tests / tools)."""
        rec = 88 if is64 else 44
        out = []
        for l in range(self._lib.fs_bla_num_levels(self._h)):
            n = self._lib.fs_bla_level_size(self._h, l)
            a = np.zeros((n, rec), np.uint8)
            err = self._lib.fs_read_bla_level(self._h, l, a.ctypes.data if n else None, n)
            if err:
                raise RuntimeError(self.ConvertErrorToString(err))
            out.append(a)
        return out

    def Render(self, algorithm, cx, cy, dx, dy, n_iterations, iteration_precision=1, T=T_F64):
        """Direct kernels.  cx = minX, cy = maxY (Fractal.cpp:1894-1915 passes the view corner).  For T_F64 the
        arguments are doubles, for T_HDR32 / T_HDR64 (mantissa, exp) pairs of the un-reduced HDRFloat values."""
        if T == T_F64:
            co = np.array([dx, dy, cx, cy], dtype=np.float64)
        else:
            co = self._pack_coords(T, [dx, dy, cx, cy])
        return self._lib.fs_render_direct(self._h, T, co.ctypes.data, int(n_iterations))

    def RenderLowPrecision(self, algorithm, coords, n_iterations, iteration_precision=1, T=T_F32):
        """Gpu1x32 (T_F32), Gpu2x32 (T_2X32), Gpu2x64 (T_2X64), Gpu4x32 (T_4X32), Gpu4x64 (T_4X64): coords from
        inputs.View.coords_direct_lp."""
        co = np.ascontiguousarray(coords)
        return self._lib.fs_render_direct_lp(self._h, T, co.ctypes.data, int(n_iterations), int(iteration_precision))

    def RenderCurrent(self, n_iterations, iter_buffer=None, color_buffer=None, reduction_results=None,
                      progressive=False):
        """iter_buffer: uint32[local_rows, rounded_width]; color_buffer: uint16[N_color_cu,4];
        reduction_results: _capi.Reduction.  Any may be None."""
        ip = iter_buffer.ctypes.data if iter_buffer is not None else None
        cp = color_buffer.ctypes.data if color_buffer is not None else None
        rp = C.addressof(reduction_results) if reduction_results is not None else None
        return self._lib.fs_render_current(self._h, int(n_iterations), ip, cp, rp, 1 if progressive else 0)

    # ---- streams
    def SyncComputeStream(self):
        return self._lib.fs_sync_compute(self._h)

    def SyncDisplayStream(self):
        return self._lib.fs_sync_display(self._h)

    def QueryComputeStream(self):
        return self._lib.fs_query_compute(self._h)

    def EnqueueComputeDoneCallback(self, fn):
        cb = _capi.DONE_CB(lambda user: fn())
        self._cbs.append(cb)
        return self._lib.fs_enqueue_done_callback(self._h, cb, None)

    def set_compressed_orbit_mode(self, runtime_decompression):
        """False / 0: SimpleCompression orbits are expanded on upload; True / 1: only the waypoints stay in HBM and the
        kernel decompresses as it walks the orbit (fs_set_compressed_orbit_mode).  Applies to the next upload."""
        return self._lib.fs_set_compressed_orbit_mode(self._h, 1 if runtime_decompression else 0)

    @property
    def orbit_device_bytes(self):
        return int(self._lib.fs_orbit_device_bytes(self._h))

    @property
    def host_fallback_bytes(self):
        """Bytes of input tables this renderer had to place in page-locked host memory (fs_host_fallback_bytes)."""
        return int(self._lib.fs_host_fallback_bytes(self._h))

    def idle_device_bytes(self):
        """Device memory this renderer keeps idle for its next allocation (fs_idle_device_bytes)."""
        return int(self._lib.fs_idle_device_bytes(self._h))

    @staticmethod
    def release_idle_device_memory(device=0):
        """Frees the idle blocks of every renderer of this process on `device` (fs_release_idle_device_memory)."""
        return int(_capi.render_lib().fs_release_idle_device_memory(int(device)))

    # ---- measurement
    def last_kernel_ms(self):
        return float(self._lib.fs_last_kernel_ms(self._h))

    def kernel_ms_history(self, n):
        """Durations (ms) of the last n iteration-kernel launches, oldest first (after SyncComputeStream)."""
        out = (C.c_float * int(n))()
        err = self._lib.fs_kernel_ms_history(self._h, out, int(n))
        if err:
            raise RuntimeError("fs_kernel_ms_history: %s" % self.ConvertErrorToString(err))
        return [float(x) for x in out]

    def kernel_ms_split_history(self, n):
        """(first, second) kernel durations of the last n launches: a two-kernel frame (HDRFloat<double> LAv2: AT pass, then the
        frame's kernel) split where the first ended; first = 0 for one-kernel frames."""
        a, b = (C.c_float * int(n))(), (C.c_float * int(n))()
        err = self._lib.fs_kernel_ms_split_history(self._h, a, b, int(n))
        if err:
            raise RuntimeError("fs_kernel_ms_split_history: %s" % self.ConvertErrorToString(err))
        return [float(x) for x in a], [float(x) for x in b]

    def set_kernel_variant(self, literal=False, lds_orbit=False, refill=False, wide_counters=False,
                           natural_tile_order=False, bla_pool=False):
        """False / 0 (default): tuned loops; True / 1: literal transcription; 2: tuned loops without the scaled runs
        (A/B references, identical results).  lds_orbit / refill: the A/B flags FS_VARIANT_LDS_ORBIT / FS_VARIANT_REFILL
        of include/fsmi355.h (orbit entries through LDS; persistent lane-refilling BLA launch); natural_tile_order:
        FS_VARIANT_NATURAL_TILE_ORDER (no "long tiles first" in the perturbation-only launch)."""
        v = (int(literal) | (VARIANT_LDS_ORBIT if lds_orbit else 0) | (VARIANT_REFILL if refill else 0) |
             (VARIANT_WIDE_COUNTERS if wide_counters else 0) |  # wide_counters: 64-bit counting kernels at any cap (tests)
             (VARIANT_NATURAL_TILE_ORDER if natural_tile_order else 0) | (0x1000 if bla_pool else 0))  # FS_VARIANT_BLA_POOL
        return self._lib.fs_set_kernel_variant(self._h, v)

    def forget_tile_costs(self):
        """The next RenderPerturbLAv2 frame of the tuned HDRFloat<float> kernel starts cold: natural tile order (it
        records costs again; the one after it runs longest tiles first)."""
        return self._lib.fs_forget_tile_costs(self._h)

    def last_frame_tile_ordered(self):
        return bool(self._lib.fs_last_frame_tile_ordered(self._h))

    def last_frame_sampled_tile_order(self):
        """The last LAv2 frame was a view's first frame with its tiles in the order of a sampled PerformAT count (round 6)."""
        return bool(self._lib.fs_last_frame_sampled_tile_order(self._h))

    def read_tile_costs(self):
        """Costs the last tuned LAv2 frame recorded, one per 8 x 8 tile of the local buffer (row-major), or None."""
        n = C.c_uint64(0)
        if self._lib.fs_read_tile_costs(self._h, None, 0, C.byref(n)) != 0:
            return None
        out = np.zeros(int(n.value), np.uint32)
        assert self._lib.fs_read_tile_costs(self._h, out.ctypes.data, out.size, C.byref(n)) == 0
        return out

    def read_tile_order(self, n_tiles):
        """Launch order of the last frame when it was an ordered one (a permutation of range(n_tiles)), else None."""
        out = np.zeros(int(n_tiles), np.uint32)
        if self._lib.fs_read_tile_order(self._h, out.ctypes.data, out.size) != 0:
            return None
        return out

    def enable_step_count(self, on=True):
        return self._lib.fs_enable_step_count(self._h, 1 if on else 0)

    def read_step_count(self):
        out = (C.c_uint64 * 8)()
        err = self._lib.fs_read_step_count(self._h, out)
        if err:
            raise RuntimeError(self.ConvertErrorToString(err))
        return {"at_iterations": out[0], "la_steps": out[1], "perturb_steps": out[2], "pixels": out[3],
                "lane_slots": out[4], "careful_steps": out[5], "scaled_steps": out[6], "scaled_runs": out[7]}

    def new_iter_buffer(self):
        dt = np.uint64 if getattr(self, "_iter_bytes", 4) == 8 else np.uint32
        return np.zeros((self.local_rows, self.rounded_width), dt)


class GPURendererGroup:
    """One frame row-tiled over several GPUs of one node behind the C ABI (fs_group_*, csrc/group.cpp): the multi-GPU
    form of GPURenderer for a single-process C++ host.  devices: HIP ordinals (repeats allowed: members then share a
    device and the gather uses peer copies -- how the one-GPU test box exercises the tiler).  transport: 0 = RCCL,
    1 = hipMemcpyPeerAsync."""

    def __init__(self, devices, transport=0):
        self._lib = _capi.render_lib()
        arr = (C.c_int * len(devices))(*devices)
        self._h = self._lib.fs_group_create(arr, len(devices), int(transport))
        if not self._h:
            raise MemoryError("fs_group_create failed")
        self._iter_bytes = 4
        self._shape = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fs_group_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def size(self):
        return self._lib.fs_group_size(self._h)

    @property
    def transport(self):
        return self._lib.fs_group_transport(self._h)

    def InitializeMemory(self, w, h, antialiasing, palInterleaved=None, palIters=0, paletteAuxDepth=0, paletteGeneration=0,
                         iter_bytes=4):
        pal_ptr = None
        if palInterleaved is not None:
            self._pal = np.ascontiguousarray(palInterleaved, dtype=np.uint16)
            pal_ptr = self._pal.ctypes.data
        self._iter_bytes = int(iter_bytes)
        self._shape = ((h + 7) // 8 * 8, (w + 15) // 16 * 16)
        return self._lib.fs_group_init_memory(self._h, w, h, antialiasing, iter_bytes, pal_ptr, palIters, paletteAuxDepth,
                                              paletteGeneration)

    def InitializePerturb(self, GenerationNumber1, Perturb1, LaReferenceHost=None, T=T_HDR32):
        err = self._lib.fs_group_upload_orbit(self._h, GenerationNumber1, T, 4, Perturb1.data_ptr, Perturb1.count,
                                              Perturb1.count, Perturb1.period)
        if err or LaReferenceHost is None:
            return err
        la = LaReferenceHost
        return self._lib.fs_group_upload_la(self._h, GenerationNumber1, T, 4, la.las_ptr, la.count, la.stages_ptr,
                                            la.stage_count, 1 if la.is_valid else 0, 1 if la.use_at else 0,
                                            C.addressof(la.at))

    def UploadBLA(self, blas, T=T_HDR32):
        if blas is None:
            return self._lib.fs_group_upload_bla(self._h, T, None, None, 0, 0)
        return self._lib.fs_group_upload_bla(self._h, T, blas.level_ptrs, blas.level_sizes, blas.num_levels, blas.lm2)

    def RenderPerturbLAv2(self, dx, dy, centerX, centerY, n_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU):
        co = GPURenderer._pack_coords(T, [dx, dy, centerX, centerY])
        return self._lib.fs_group_render_lav2(self._h, T, Mode, parity, co.ctypes.data, int(n_iterations))

    def RenderPerturbBLA(self, dx, dy, centerX, centerY, n_iterations, T=T_HDR32):
        co = GPURenderer._pack_coords(T, [dx, dy, centerX, centerY])
        return self._lib.fs_group_render_bla(self._h, T, co.ctypes.data, int(n_iterations))

    def Render(self, cx, cy, dx, dy, n_iterations, T=T_F64):
        """Direct kernels over the row-tiled frame (GPURenderer.Render: cx = minX, cy = maxY)."""
        co = np.array([dx, dy, cx, cy], dtype=np.float64) if T == T_F64 else GPURenderer._pack_coords(T, [dx, dy, cx, cy])
        return self._lib.fs_group_render_direct(self._h, T, co.ctypes.data, int(n_iterations))

    def ClearMemory(self):
        return self._lib.fs_group_clear(self._h)

    def RenderCurrent(self, n_iterations, iter_buffer=None, reduction_results=None, color_buffer=None, progressive=False):
        """GPURenderer::RenderCurrent(n, iters, colors, reduction, progressive) for the row-tiled frame.  progressive: a
        snapshot on the display streams (SyncDisplay waits for it)."""
        ip = iter_buffer.ctypes.data if iter_buffer is not None else None
        rp = C.addressof(reduction_results) if reduction_results is not None else None
        cp = color_buffer.ctypes.data if color_buffer is not None else None
        if cp is None and not progressive:
            return self._lib.fs_group_render_current(self._h, int(n_iterations), ip, rp)
        return self._lib.fs_group_render_current_colors(self._h, int(n_iterations), ip, cp, rp, 1 if progressive else 0)

    def new_color_buffer(self):
        """uint16[N_color_cu, 4]: the Color16 buffer RenderCurrent fills (padded to 16 x 8 blocks of colour pixels)."""
        n = self._lib.fs_color_buffer_elements(self._lib.fs_group_renderer(self._h, 0))
        return np.zeros((int(n), 4), np.uint16)

    def Sync(self):
        return self._lib.fs_group_sync(self._h)

    def SyncDisplay(self):
        return self._lib.fs_group_sync_display(self._h)

    def SetHostPath(self, direct):
        """fs_group_set_host_path: False = gather through device 0, True = every member copies its own bands to the host."""
        return self._lib.fs_group_set_host_path(self._h, 1 if direct else 0)

    def WaitCurrent(self, frames_back=0):
        """Host waits for the RenderCurrent issued `frames_back` calls ago (0 = latest, 1 = the one before): two frames may
        be in flight (render k+1 while frame k is gathered and copied out)."""
        return self._lib.fs_group_wait_current(self._h, int(frames_back))

    def renderer(self, rank):
        """Borrowed GPURenderer view of member `rank` (fs_group_renderer): measurement hooks only."""
        m = GPURenderer.__new__(GPURenderer)
        m._lib = self._lib
        m._h = self._lib.fs_group_renderer(self._h, int(rank))
        m._borrowed = True
        return m

    def gather_ms(self):
        return float(self._lib.fs_group_gather_ms(self._h))

    def new_iter_buffer(self):
        return np.zeros(self._shape, np.uint64 if self._iter_bytes == 8 else np.uint32)
