"""CPU-only: pin the oracle (and the host input builders it is fed by) against the reference's OWN golden
vectors -- the CRC-64s of the PNG files in FractalSharkTest/TestRenderGoldens.cpp:84-97 -- and against the
committed fixtures.  A matching CRC pins the whole chain view -> GMP orbit -> LA/BLA table -> CPU render ->
palette -> PNG bit-exactly."""
import os

import numpy as np
import pytest

import _oracle
from fractalshark_amd import inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
W = H = 256  # kGoldenWidth / kGoldenHeight

needs_pin = pytest.mark.skipif(_oracle.pin_lib() is None,
                               reason="oracle/_ref/libpngpin.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def view5(native_libs):
    v = inputs.View.builtin(5, W, H)
    ob = inputs.Orbit(v)
    return v, ob


@needs_pin
@pytest.mark.parametrize("aa,expected", [(1, "1275500d639ad02e"), (4, "39671027bacf2567")])
def test_golden_view0_cpu64(native_libs, aa, expected):
    v = inputs.View.builtin(0, W, H, antialiasing=aa)
    it = _oracle.direct_f64(v, aa=aa)
    assert _oracle.png_crc64(it, W, H, aa, v.num_iterations) == expected


@needs_pin
def test_golden_view5_cpu32_perturbed_blav2_hdr(view5):
    v, ob = view5
    la = inputs.LATable(ob)
    it = _oracle.lav2_hdr32(v, ob, la, stage_test=0)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "1233a56b293e7b08"


@needs_pin
def test_golden_view5_cpu32_perturbed_bla_hdr(view5):
    v, ob = view5
    bla = inputs.BLATable(ob)
    it = _oracle.bla_hdr32(v, ob, bla)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "634d826801d54979"


def test_view5_inputs_match_fixture(native_libs):
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    assert [ob.count, ob.period] == list(GOLD["view5_orbit_count_period"])
    assert ob.entries()[:64].tobytes() == GOLD["view5_orbit_head"].tobytes()
    assert ob.entries()[-64:].tobytes() == GOLD["view5_orbit_tail"].tobytes()
    assert [la.count, la.stage_count, int(la.use_at)] == list(GOLD["view5_la_count_stages"])
    assert la.stages().tobytes() == GOLD["view5_la_stages"].tobytes()
    assert la.records()[:32].tobytes() == GOLD["view5_la_head"].tobytes()
    assert v.coords_perturb_hdr32(ob).tobytes() == GOLD["view5_coords_hdr32_64x36"].tobytes()
    # entry 0 is the explicit zero entry with exponent INT32_MIN >> 3
    e0 = ob.entries()[0]
    assert (e0["mx"], e0["ex"], e0["ey"], e0["my"]) == (0.0, -268435456, -268435456, 0.0)


def test_oracle_reproduces_fixtures(native_libs):
    v0 = inputs.View.builtin(0, 64, 48)
    assert np.array_equal(_oracle.direct_f64(v0), GOLD["view0_direct_f64_64x48"])
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    bla = inputs.BLATable(ob)
    assert np.array_equal(_oracle.lav2_hdr32(v, ob, la, stage_test=0), GOLD["view5_lav2_cpu_64x36"])
    assert np.array_equal(_oracle.lav2_hdr32(v, ob, la, stage_test=1), GOLD["view5_lav2_gpustage_64x36"])
    assert np.array_equal(_oracle.bla_hdr32(v, ob, None), GOLD["view5_po_64x36"])
    assert np.array_equal(_oracle.bla_hdr32(v, ob, bla), GOLD["view5_bla_64x36"])


def test_oracle_thread_count_independent(native_libs):
    """Row claiming must not change results (Fractal.cpp:2523-2543)."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    a = _oracle.lav2_hdr32(v, ob, la, threads=1, stage_test=1)
    b = _oracle.lav2_hdr32(v, ob, la, threads=5, stage_test=1)
    assert np.array_equal(a, b)


def test_la_table_threading_replay_equivalent(native_libs):
    """The reference builds stage 0 of the LA table on min(count/50000, cores) threads (LAReference.cpp:236-251);
    View 5's orbit (16046 entries) is below the threshold, so every host_threads value gives the same table."""
    v = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v)
    assert ob.count < 50000
    a = inputs.LATable(ob, host_threads=1).records().tobytes()
    b = inputs.LATable(ob, host_threads=8).records().tobytes()
    assert a == b


# ---- HDRFloat<double> family and the HDR direct functions: five more of the reference's golden CRCs
@needs_pin
def test_golden_view5_cpu64_perturbed_blav2_hdr(native_libs):
    v = inputs.View.builtin(5, W, H)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    it = _oracle.lav2_hdr32(v, ob, la, stage_test=0)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "ca7ad7c5f9cf750e"


@needs_pin
def test_golden_view5_cpu64_perturbed_bla_hdr(native_libs):
    v = inputs.View.builtin(5, W, H)
    ob = inputs.Orbit(v, is64=True)
    bla = inputs.BLATable(ob)
    it = _oracle.bla_hdr32(v, ob, bla)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "c91e33c3eb85b33d"


@needs_pin
def test_golden_view1_cpu64_perturbed_bla_hdr(native_libs):
    v = inputs.View.builtin(1, W, H)
    ob = inputs.Orbit(v, is64=True)
    bla = inputs.BLATable(ob)
    it = _oracle.bla_hdr32(v, ob, bla)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "d0c8921c878f6dc3"


@needs_pin
@pytest.mark.parametrize("is64,expected", [(False, "66ba2caaaa7f8013"), (True, "1275500d639ad02e")])
def test_golden_view0_cpuhdr(native_libs, is64, expected):
    v = inputs.View.builtin(0, W, H)
    it = _oracle.direct_hdr(v, is64)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == expected


# ---- PerturbExtras::SimpleCompression ("RC" algorithms): compressed reference orbit + runtime decompression
@needs_pin
@pytest.mark.parametrize("is64,expected", [(False, "b956600cfdfe431a"), (True, "68df9ceecaf1a667")])
def test_golden_view5_rc_blav2_hdr(native_libs, is64, expected):
    v = inputs.View.builtin(5, W, H)
    ob = inputs.Orbit(v, is64=is64, compression_exp=20)
    assert ob.compressed_count < ob.count
    la = inputs.LATable(ob)
    it = _oracle.lav2_hdr32(v, ob, la, stage_test=0)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == expected


@needs_pin
def test_golden_view5_cpu64_perturbed_bla_plain_double(native_libs):
    """The twelfth golden: plain-double perturbation + BLA (Cpu64PerturbedBLA)."""
    v = inputs.View.builtin(5, W, H)
    ob = inputs.OrbitF64(v)
    it = _oracle.bla_f64(v, ob)
    assert _oracle.png_crc64(it, W, H, 1, v.num_iterations) == "f201db00ade569fc"
