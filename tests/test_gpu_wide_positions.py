"""GPU: IterType-wide orbit POSITIONS for waypoint-resident (PerturbExtras::SimpleCompression) orbits.

The reference templates every orbit index on IterType (Perturb.cuh:21-23,202-203,247-271; LAInfoI.h:5-19): that is what
lets a compressed orbit of 2^32 and more uncompressed entries run -- only its waypoints are resident.  Here
fs_set_compressed_orbit_mode(1) + the 64-bit counting instantiation walk the orbit with 64-bit positions and read the
reference's uint64_t LA records as they are.  Checked three ways:
  * the two RC golden cases (the reference's own CRC-64 literals) rendered through that instantiation, forced at the
    view's own cap (FS_VARIANT_WIDE_COUNTERS), uint64_t tables uploaded and KEPT in the uint64_t layout;
  * the decompression cursor on waypoint lists whose indices straddle 2^32, against the same list shifted below 2^32
    (the value at an index is a pure function of the waypoint values and of the distances between them);
  * uploads that used to be refused for their size are accepted, and a render that cannot serve them says so."""
import ctypes as C

import numpy as np
import pytest

import _oracle
import golden_cases as gc
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR32, T_HDR64, _capi, inputs

pytestmark = pytest.mark.gpu
UNSUPPORTED = 10100


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.set_compressed_orbit_mode(False)
    r.close()


RC_CASES = [c for c in gc.CASES if "RC" in c[2]]


@pytest.mark.parametrize("name,view_n,alg,aa,crc64", RC_CASES, ids=[c[0] + "-64-bit-positions" for c in RC_CASES])
def test_rc_goldens_through_the_64_bit_position_instantiation(renderer, native_libs, name, view_n, alg, aa, crc64):
    assert len(RC_CASES) == 2
    v, ob, table = gc.build_inputs(inputs, view_n, alg, aa)
    assert ob.compressed
    T = T_HDR64 if ob.is64 else T_HDR32
    la64 = inputs.LATableU64(table)
    r = renderer
    n = v.num_iterations
    try:
        assert r.set_compressed_orbit_mode(True) == 0
        assert r.set_kernel_variant(0, wide_counters=True) == 0
        assert r.InitializeMemory(gc.W * aa, gc.H * aa, aa, None, 0, 0, 0, False, iter_bytes=8) == 0
        assert r.InitializePerturb(7, ob, 0, None, la64, iter_bytes=8) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob)), n, T=T, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
        it = r.new_iter_buffer()
        assert it.dtype == np.uint64
        assert r.RenderCurrent(n, it) == 0
        assert r.SyncComputeStream() == 0
        # the table is resident in the uint64_t layout: a kernel that reads an expanded orbit cannot use it
        assert r.set_compressed_orbit_mode(False) == 0
        ob_plain = inputs.Orbit(v, is64=ob.is64)
        assert r.InitializePerturb(8, ob_plain) == 0  # (orbit only: the table of generation 7 stays)
        assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob_plain)), n, T=T, Mode=LAV2_FULL,
                                   parity=PARITY_CPU) == UNSUPPORTED
    finally:
        r.set_kernel_variant(0)
        r.set_compressed_orbit_mode(False)
    assert int(it.max()) < (1 << 32)
    it32 = it.astype(np.uint32)
    assert _oracle.pin_lib() is not None
    assert _oracle.png_crc64(it32, gc.W, gc.H, aa, n) == crc64, name


def _waypoints(is64, indices, seed):
    """A synthetic waypoint list (reference layout fs_orbit_hdr32_rc / fs_orbit_hdr64_rc) with the given orbit indices."""
    rng = np.random.default_rng(seed)
    if is64:   # fs_orbit_hdr64_rc, 40 B
        dt = np.dtype([("idx", "<u8"), ("mx", "<f8"), ("ex", "<i4"), ("p0", "<i4"), ("ey", "<i4"), ("p1", "<i4"), ("my", "<f8")])
    else:      # fs_orbit_hdr32_rc, 24 B
        dt = np.dtype([("idx", "<u8"), ("mx", "<f4"), ("ex", "<i4"), ("ey", "<i4"), ("my", "<f4")])
    assert dt.itemsize == (40 if is64 else 24)
    wp = np.zeros(len(indices), dt)
    wp["mx"] = rng.uniform(1.0, 1.9, len(indices)) * rng.choice([-1, 1], len(indices))
    wp["my"] = rng.uniform(1.0, 1.9, len(indices)) * rng.choice([-1, 1], len(indices))
    wp["ex"] = rng.integers(-3, 0, len(indices))
    wp["ey"] = rng.integers(-3, 0, len(indices))
    wp["idx"] = np.asarray(indices, np.uint64)
    wp["mx"][0] = wp["my"][0] = 0.0  # the orbit starts at zero
    return wp


@pytest.mark.parametrize("is64", [False, True])
def test_cursor_with_indices_straddling_2_to_32(renderer, native_libs, is64):
    r = renderer
    lib = r._lib
    T = T_HDR64 if is64 else T_HDR32
    base = (1 << 32) - 40
    big = [0, 100, base + 10, base + 37, base + 40 + 9, base + 40 + 300, (1 << 33) + 5]   # straddles 2^32 twice over
    shift = base - 1000
    small = [0, 100] + [i - shift for i in big[2:]]
    cdt = np.dtype([("re", "<f8"), ("im", "<f8"), ("e", "<i4"), ("pad", "<i4")]) if is64 else \
        np.dtype([("re", "<f4"), ("im", "<f4"), ("e", "<i4")])
    low = np.zeros(2, np.dtype([("m", "<f8"), ("e", "<i4"), ("pad", "<i4")]) if is64 else np.dtype([("m", "<f4"), ("e", "<i4")]))
    low["m"] = [-1.25, 1.5]
    low["e"] = [-1, -3]

    def probe(indices, wide, start, n):
        wp = _waypoints(is64, indices, 7)
        assert lib.fs_upload_orbit_compressed(r._h, 0, T, 8, wp.ctypes.data, len(wp), int(indices[-1]) + 50, 0,
                                              low[0:1].ctypes.data, low[1:2].ctypes.data) == 0
        out = np.zeros(n, cdt)
        assert lib.fs_seq_cursor_probe(r._h, 1 if wide else 0, int(start), n, out.ctypes.data) == 0
        return out

    try:
        assert r.set_compressed_orbit_mode(True) == 0
        assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
        # (every start lies a few hundred entries at most behind a waypoint: the cursor reaches it by iterating from there)
        for start_big, n in ((base + 10, 80), (base + 12, 70), ((1 << 32) - 2, 12), ((1 << 32) + 1, 60), (base + 40 + 250, 120)):
            got = probe(big, True, start_big, n)
            ref = probe(small, False, start_big - shift, n)     # 32-bit positions, the same distances
            ref64 = probe(small, True, start_big - shift, n)    # ... and the 64-bit cursor on small indices
            for f in ("re", "im", "e"):
                assert np.array_equal(got[f], ref[f]), (start_big, f)
                assert np.array_equal(ref64[f], ref[f]), (start_big, f)
        # the values are not trivially equal: a waypoint is loaded exactly where its index comes up
        a = probe(big, True, base + 36, 3)
        wp = _waypoints(is64, big, 7)
        e3 = max(int(wp["ex"][3]), int(wp["ey"][3]))
        assert int(a["e"][1]) == e3 and float(a["re"][1]) == float(wp["mx"][3]) * 2.0 ** (int(wp["ex"][3]) - e3)
        # an upload that used to be refused for its size alone: 2^33 uncompressed entries, a period beyond 2^32
        wpb = _waypoints(is64, big, 3)
        assert lib.fs_upload_orbit_compressed(r._h, 0, T, 8, wpb.ctypes.data, len(wpb), (1 << 33) + 77, (1 << 32) + 9,
                                              low[0:1].ctypes.data, low[1:2].ctypes.data) == 0
        # ... expanded mode cannot hold it and says so
        assert r.set_compressed_orbit_mode(False) == 0
        assert lib.fs_upload_orbit_compressed(r._h, 0, T, 8, wpb.ctypes.data, len(wpb), (1 << 33) + 77, 0,
                                              low[0:1].ctypes.data, low[1:2].ctypes.data) == UNSUPPORTED
    finally:
        r.set_compressed_orbit_mode(False)


@pytest.mark.parametrize("is64", [False, True])
def test_la_table_with_step_lengths_beyond_32_bits_is_accepted(renderer, native_libs, is64):
    """fs_upload_la with a uint64_t table one of whose step lengths does not fit 32 bits: kept in the uint64_t layout under
    the waypoint-resident mode (no size refusal), refused by the modes that read an expanded orbit."""
    r = renderer
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v, is64=is64, compression_exp=20)
    la64 = inputs.LATableU64(inputs.LATable(ob))
    T = T_HDR64 if is64 else T_HDR32
    rec = la64._las
    saved = rec[0].copy()
    rec[0, -16:-8] = np.frombuffer(np.uint64((1 << 32) + 5).tobytes(), np.uint8)  # record 0's StepLength
    try:
        assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
        assert r.set_compressed_orbit_mode(False) == 0
        assert r.InitializePerturb(0, ob, 0, None, la64, iter_bytes=8) == UNSUPPORTED
        assert r.set_compressed_orbit_mode(True) == 0
        assert r.InitializePerturb(0, ob, 0, None, la64, iter_bytes=8) == 0
        # the frame renders: record 0 (the first stage's first record) is never usable with a step beyond the cap
        assert r.ClearMemory() == 0
        assert r.RenderPerturbLAv2(None, None, None, *_pairs(v.coords_perturb(ob)), v.num_iterations, T=T, Mode=LAV2_FULL,
                                   parity=PARITY_CPU_GPUSTAGE) == 0
        assert r.SyncComputeStream() == 0
    finally:
        rec[0] = saved
        r.set_compressed_orbit_mode(False)


def test_waypoint_resident_mode_for_the_plain_types_and_hdr2x32(renderer, native_libs):
    """fs_set_compressed_orbit_mode(1) for float / double / CudaDblflt / HDRFloat<CudaDblflt> (GPU_Render.cu:518-523,532-537
    instantiates SimpleCompression for them too): only the waypoints are resident and the LAv2 kernels walk them with a cursor
    per pixel.  Same frames as the expanding mode (whose kernels are checked against the restated CUDA kernels), all modes;
    the resident orbit is the waypoints."""
    from fractalshark_amd import LAV2_PO, T_2X32, T_F32, T_F64, T_HDR2X32
    from test_plain_oracle import shallow_view
    r = renderer
    vs = shallow_view("1e-12")
    w, h = 70, 37
    try:
        for kind, T, rc_bytes in (("f32", T_F32, 16), ("f64", T_F64, 24), ("2x32", T_2X32, 24)):
            pin = inputs.PlainInputs(vs, kind, compression_exp=20)
            assert pin.compressed and 1 < pin.compressed_count < pin.count
            frames = {}
            for seq in (False, True):
                assert r.set_compressed_orbit_mode(seq) == 0
                assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
                assert r.InitializePerturbPlain(0, pin) == 0
                if seq:
                    assert r.orbit_device_bytes == pin.compressed_count * rc_bytes
                for mode in (LAV2_FULL, LAV2_PO, LAV2_LAO):
                    assert r.ClearMemory() == 0
                    assert r.RenderPerturbLAv2Plain(pin, vs.num_iterations, Mode=mode) == 0
                    out = r.new_iter_buffer()
                    assert r.RenderCurrent(vs.num_iterations, out) == 0
                    assert r.SyncComputeStream() == 0
                    frames[(seq, mode)] = out
            for mode in (LAV2_FULL, LAV2_PO, LAV2_LAO):
                assert np.array_equal(frames[(True, mode)], frames[(False, mode)]), (kind, mode)
            assert frames[(True, LAV2_FULL)].max() > 0
        # HDRFloat<CudaDblflt>
        v = inputs.View.builtin(5, 64, 36)
        o = inputs.Orbit(v, is64=True, compression_exp=20)
        la = inputs.LATable(o, use_small_exponents=True)
        o2, la2 = inputs.Orbit2x32(o), inputs.LATable2x32(la)
        tr = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in v.coords_perturb_2x32(o2)]
        frames = {}
        for seq in (False, True):
            assert r.set_compressed_orbit_mode(seq) == 0
            assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False) == 0
            assert r.InitializePerturb(0, o2, 0, None, la2) == 0
            if seq:
                assert r.orbit_device_bytes == o2.compressed_count * 32
            for mode in (LAV2_FULL, LAV2_LAO):
                assert r.ClearMemory() == 0
                assert r.RenderPerturbLAv2(None, None, None, *tr, v.num_iterations, T=T_HDR2X32, Mode=mode) == 0
                out = r.new_iter_buffer()
                assert r.RenderCurrent(v.num_iterations, out) == 0
                assert r.SyncComputeStream() == 0
                frames[(seq, mode)] = out
        for mode in (LAV2_FULL, LAV2_LAO):
            assert np.array_equal(frames[(True, mode)], frames[(False, mode)]), mode
    finally:
        r.set_compressed_orbit_mode(False)
