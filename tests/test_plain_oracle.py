"""CPU-only checks of the non-HDR LAv2 chain (T = float / double / CudaDblflt: Gpu1x32 / Gpu1x64 / Gpu2x32
PerturbedLAv2*, the algorithms Fractal's AUTO mode uses between zoom 1e4 and 1e34, Fractal.cpp:958-966).

The reference has no CPU RenderAlgorithm for LAv2 on a plain type, so oracle/gpu_ref_plain.cpp is "parity unpinned".
What can be pinned without the CUDA reference is pinned here:
  * the plain-double orbit and LA table against the golden-pinned HDRFloat<double> ones, value for value;
  * the plain-double perturbation loop against the pinned CPU function (Cpu64PerturbedBLAV2HDR's per-pixel loop), which
    on shallow views runs perturbation only (its zero-with-exponent-0 start value makes every LA step unusable);
  * float / CudaDblflt renders against the double render (same algorithm, fewer mantissa bits).
"""
from decimal import Decimal, getcontext

import numpy as np
import pytest

import _oracle
from fractalshark_amd import inputs


def shallow_view(width, n_iter=20000, W=64, H=36):
    """A view of the given width (decimal string) centred on View 5's centre."""
    getcontext().prec = 60
    cx = Decimal("-0.5482057480704757084582125675467330293766992786373239")
    cy = Decimal("-0.5775708389036038428051089822018505586755517268027721")
    w = Decimal(width)
    h = w * H / W
    return inputs.View(str(cx - w / 2), str(cy - h / 2), str(cx + w / 2), str(cy + h / 2), W, H,
                       num_iterations=n_iter)


_R = np.dtype([("m", "<f8"), ("e", "<i4"), ("p", "<i4")])
_C = np.dtype([("re", "<f8"), ("im", "<f8"), ("e", "<i4"), ("p", "<i4")])
_LA64 = np.dtype([("Ref", _C), ("ZCoeff", _C), ("CCoeff", _C), ("LAThreshold", _R), ("LAThresholdC", _R),
                  ("MinMag", _R), ("StepLength", "<u4"), ("NextStageLAIndex", "<u4")])


@pytest.mark.parametrize("width", ["1e-6", "1e-12", "1e-20", "1e-28"])
def test_plain_double_inputs_equal_the_pinned_hdr64_inputs(native_libs, width):
    v = shallow_view(width)
    p = inputs.PlainInputs(v, "f64")
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o)
    e, po = o.entries(), p.orbit()
    assert p.count == o.count and p.period == o.period
    assert np.array_equal(np.ldexp(e["mx"], e["ex"]), po["x"]) and np.array_equal(np.ldexp(e["my"], e["ey"]), po["y"])
    r, pl = la.records().view(_LA64).reshape(-1), p.las()
    assert la.count == p.la_count and la.stage_count == p.stage_count and la.use_at == p.use_at
    for f in ("Ref", "ZCoeff", "CCoeff"):
        assert np.array_equal(np.ldexp(r[f]["re"], r[f]["e"]), pl[f]["re"]), f
        assert np.array_equal(np.ldexp(r[f]["im"], r[f]["e"]), pl[f]["im"]), f
    for f in ("LAThreshold", "LAThresholdC", "MinMag"):
        assert np.array_equal(np.ldexp(r[f]["m"], r[f]["e"]), pl[f]), f
    assert np.array_equal(r["StepLength"], pl["StepLength"])
    assert np.array_equal(r["NextStageLAIndex"], pl["NextStageLAIndex"])
    assert np.array_equal(la.stages().view(np.uint32).reshape(-1), p.stages().view(np.uint32).reshape(-1))


def test_record_layouts():
    for kind, sizes in (("f32", (8, 44, 72)), ("f64", (16, 80, 144)), ("2x32", (16, 80, 140))):
        o, la, at, _ = inputs._plain_dtypes(kind)
        assert (o.itemsize, la.itemsize, at.itemsize) == sizes


def test_2x32_conversion_is_exact_to_48_bits(native_libs):
    v = shallow_view("1e-12")
    p64, p2 = inputs.PlainInputs(v, "f64"), inputs.PlainInputs(v, "2x32")
    a, b = p64.orbit(), p2.orbit()
    h, t = b["x_head"].astype(np.float64), b["x_tail"].astype(np.float64)
    assert (np.abs(h + t - a["x"]) <= np.abs(a["x"]) * 2.0 ** -47).all()
    assert (np.float32(h + t) == b["x_head"]).all()  # normalised
    assert np.array_equal(p64.las()["StepLength"], p2.las()["StepLength"])
    c64, c2 = p64.coords(), p2.coords()
    assert np.allclose(c2["head"].astype(np.float64) + c2["tail"], c64, rtol=2.0 ** -46, atol=0)


@pytest.mark.parametrize("width", ["1e-6", "1e-12", "1e-20", "1e-28"])
def test_plain_double_perturbation_loop_equals_the_pinned_cpu_function(native_libs, width):
    v = shallow_view(width)
    p = inputs.PlainInputs(v, "f64")
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o)
    ref, st = _oracle.lav2_hdr32(v, o, la, stage_test=1, stats=True)
    po, sp = _oracle.gpu_lav2_plain(v, p, mode=1, stats=True)
    assert sp["at_iterations"] == 0 and sp["la_steps"] == 0
    if st["la_steps"] <= 20:  # the CPU function ran (almost) perturbation only
        assert (po[:36, :64] == ref[:36, :64]).mean() > 0.995
    full, sf = _oracle.gpu_lav2_plain(v, p, mode=0, stats=True)
    lao = _oracle.gpu_lav2_plain(v, p, mode=2)
    assert sf["la_steps"] > 0 and sf["perturb_steps"] < sp["perturb_steps"]
    d = full[:36, :64].astype(np.int64) - po[:36, :64]
    assert (np.abs(d) <= 2).mean() > 0.9, np.unique(d, return_counts=True)
    assert (lao[:36, :64] <= full[:36, :64]).all()


def test_float_and_2x32_track_the_double_render(native_libs):
    v = shallow_view("1e-12")
    r64 = _oracle.gpu_lav2_plain(v, inputs.PlainInputs(v, "f64"), mode=1)[:36, :64].astype(np.int64)
    r2 = _oracle.gpu_lav2_plain(v, inputs.PlainInputs(v, "2x32"), mode=1)[:36, :64].astype(np.int64)
    r32 = _oracle.gpu_lav2_plain(v, inputs.PlainInputs(v, "f32"), mode=1)[:36, :64].astype(np.int64)
    assert (np.abs(r2 - r64) <= 2).mean() > 0.98
    assert (np.abs(r32 - r64) <= 2).mean() > 0.75


def test_2x32_at_validity_uses_the_reference_operator_as_written(native_libs):
    """CudaDblflt's operator<= is `!(b > a)` (CudaDblflt.h:218-222), so ATInfo::isValid (`cheb(dc) <= ThresholdC`)
    accepts the pixels *outside* the AT radius in the 2x32 kernel.  The oracle restates it as written: on a view whose
    double kernel uses AT for a handful of pixels, the 2x32 kernel uses it for (nearly) all of them."""
    v = shallow_view("1e-20")
    p64, p2 = inputs.PlainInputs(v, "f64"), inputs.PlainInputs(v, "2x32")
    assert p64.use_at and p2.use_at
    _, s64 = _oracle.gpu_lav2_plain(v, p64, mode=2, stats=True)
    _, s2 = _oracle.gpu_lav2_plain(v, p2, mode=2, stats=True)
    assert s64["at_iterations"] <= 4 and s2["at_iterations"] >= 2000


# ---- PerturbExtras::SimpleCompression for the non-HDR types (Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedRCLAv2*)
@pytest.mark.parametrize("width", ["1e-6", "1e-12", "1e-20"])
def test_plain_double_compression_equals_the_pinned_hdr64_compression(native_libs, width):
    """The HDRFloat<double> compressor / RuntimeDecompressor chain is pinned by a golden CRC (Cpu64PerturbedRCBLAV2HDR);
    in binary64 range the plain-double chain must pick the same waypoints and rebuild the same orbit."""
    v = shallow_view(width)
    p = inputs.PlainInputs(v, "f64", compression_exp=20)
    o = inputs.Orbit(v, is64=True, compression_exp=20)
    assert p.compressed and p.count == o.count and p.period == o.period
    assert 1 < p.compressed_count == o.compressed_count < p.count
    wp = p.waypoints()
    assert wp["index"][0] == 0 and wp["x"][0] == 0 and wp["y"][0] == 0 and (np.diff(wp["index"].astype(np.int64)) > 0).all()
    e, po = o.entries(), p.orbit()
    assert np.array_equal(np.ldexp(e["mx"], e["ex"]), po["x"]) and np.array_equal(np.ldexp(e["my"], e["ey"]), po["y"])
    # waypoints are orbit entries; everything between them is rebuilt
    assert np.array_equal(po["x"][wp["index"]], wp["x"]) and np.array_equal(po["y"][wp["index"]], wp["y"])
    u = inputs.PlainInputs(v, "f64").orbit()
    assert np.array_equal(u["x"][wp["index"]], wp["x"])
    err2 = (po["x"] - u["x"]) ** 2 + (po["y"] - u["y"]) ** 2
    assert (err2[1:] * 1e20 <= (u["x"] ** 2 + u["y"] ** 2)[1:] * 1.001).all()  # the compressor's own bound
    assert (err2 > 0).any()


def test_plain_float_compression_bound_and_table(native_libs):
    v = shallow_view("1e-6")
    # binary32 carries ~7 digits: an error bound of 10^-(exp/2) = 1e-5 relative is the tightest that still compresses
    p = inputs.PlainInputs(v, "f32", compression_exp=10)
    u = inputs.PlainInputs(v, "f32")
    assert 1 < p.compressed_count < p.count == u.count
    po, uo, wp = p.orbit(), u.orbit(), p.waypoints()
    assert wp.dtype.itemsize == 16
    assert np.array_equal(uo["x"][wp["index"]], wp["x"]) and np.array_equal(po["y"][wp["index"]], wp["y"])
    err2 = (po["x"].astype(np.float64) - uo["x"]) ** 2 + (po["y"].astype(np.float64) - uo["y"]) ** 2
    assert (err2[1:] * 1e10 <= (uo["x"].astype(np.float64) ** 2 + uo["y"].astype(np.float64) ** 2)[1:] * 1.01).all()
    # LAReference over a SimpleCompression orbit uses period divisor 8 instead of 2 (LAReference.cpp:12-19)
    assert p.is_valid and p.la_count > 0
    low = p.orbit_low()
    assert low.dtype == np.float32 and abs(float(low[0]) + 0.5482057) < 1e-6


def test_2x32_compressed_orbit_is_rebuilt_in_2x32_arithmetic(native_libs):
    """CudaDblflt waypoints are the converted binary64 ones (CopyFullOrbitVector); the GPU-side rebuild runs in
    double-float arithmetic, so it tracks -- but is not -- the converted binary64 rebuild."""
    v = shallow_view("1e-12")
    p2 = inputs.PlainInputs(v, "2x32", compression_exp=20)
    p64 = inputs.PlainInputs(v, "f64", compression_exp=20)
    wp2, wp64 = p2.waypoints(), p64.waypoints()
    assert wp2.dtype.itemsize == 24 and np.array_equal(wp2["index"], wp64["index"])
    assert np.array_equal(np.float32(wp64["x"]), wp2["x_head"])
    full = _oracle.decompress_p2x32(p2)
    assert np.array_equal(full["x_head"][wp2["index"]], wp2["x_head"])
    assert np.array_equal(full["y_tail"][wp2["index"]], wp2["y_tail"])
    x = full["x_head"].astype(np.float64) + full["x_tail"]
    ref = p64.orbit()["x"]
    big = np.abs(ref) > 1e-3
    rel = np.abs(x - ref)[big] / np.abs(ref)[big]
    assert rel.max() < 1e-8 and (x != ref).any()


def test_hdr2x32_compressed_orbit_is_rebuilt_in_2x32_arithmetic(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    o = inputs.Orbit(v, is64=True, compression_exp=20)
    o2 = inputs.Orbit2x32(o)
    assert o2.compressed and o2.compressed_count == o.compressed_count
    wp = o2.waypoints()
    assert wp.dtype.itemsize == 32 and wp["index"][0] == 0
    low = o2.orbit_low()
    # HDRFloat<CudaDblflt>(double): mantissa in [1, 2) (HDRFloat.h:341-349)
    assert 1.0 <= abs(float(low["head"][0])) < 2.0 and int(low["e"][0]) == -1
    assert abs((float(low["head"][0]) + float(low["tail"][0])) * 0.5 + 0.548205748070475708) < 1e-13
    full = _oracle.decompress_hdr2x32(o2)
    assert np.array_equal(full["x_head"][wp["index"]], wp["x_head"]) and np.array_equal(full["ey"][wp["index"]], wp["ey"])
    e = o.entries()
    x = np.ldexp(full["x_head"].astype(np.float64) + full["x_tail"], full["ex"])
    ref = np.ldexp(e["mx"], e["ex"])
    big = np.abs(ref) > 1e-3
    assert (np.abs(x - ref)[big] / np.abs(ref)[big]).max() < 1e-8
