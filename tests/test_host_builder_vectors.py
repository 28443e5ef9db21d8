"""Host-side inputs of the hot path against values the reference's OWN unit tests hold (tests/golden/host_builder_vectors.json,
extracted as data by tests/golden/make_host_builder_vectors.py): the LAParameters defaults every LAv2 table is built with
(TestLAParameters.cpp) and the MPIR precision a view is given (TestPrecisionCalculator.cpp: the larger binary exponent of the
view's width / height, in GMP's [0.5, 1) convention, + 120 bits) -- the number that decides every bit of the reference orbit."""
import ctypes as C
import json
import os
from decimal import Decimal, getcontext

import pytest

from fractalshark_amd import _capi, inputs

VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_builder_vectors.json")))
ORDER = ["GetDetectionMethod", "GetLAThresholdScaleExp", "GetLAThresholdCScaleExp", "GetStage0PeriodDetectionThreshold2Exp",
         "GetPeriodDetectionThreshold2Exp", "GetStage0PeriodDetectionThresholdExp", "GetPeriodDetectionThresholdExp"]


def test_la_parameter_defaults(native_libs):
    out = (C.c_int32 * 7)()
    _capi.inputs_lib().fsh_la_default_params(out)
    want = VEC["la_parameters"]["defaults"]
    assert set(want) == set(ORDER)
    assert list(out) == [want[k] for k in ORDER]


@pytest.mark.parametrize("case", VEC["precision"], ids=[c["name"] for c in VEC["precision"]])
def test_view_precision(native_libs, case):
    if "box" in case:
        x0, y0, x1, y1 = case["box"]
    else:  # a square box of half-width 2^e around the origin, written out exactly
        getcontext().prec = 80
        h = Decimal(2) ** case["box_pow2"]["half_width_exp"]
        x0, y0, x1, y1 = str(-h), str(-h), str(h), str(h)
    v = inputs.View(x0, y0, x1, y1, 64, 64, num_iterations=100)
    assert v.precision_bits == case["precision_bits"], case["source"]


@pytest.mark.parametrize("case", VEC["double_float_split"], ids=[c["name"] for c in VEC["double_float_split"]])
def test_double_to_double_float_split(native_libs, case):
    """TestCudaDblflt.cpp: what MattDblflt(double) / CudaDblflt(double) must give back.  The product's converter is the one every
    HDRFloat<CudaDblflt> / CudaDblflt orbit, table and coordinate goes through (fsh_convert_*_to_*2x32, host/refinputs.cpp)."""
    import numpy as np
    lib = _capi.inputs_lib()
    src = np.zeros(1, np.dtype([("x", "<f8"), ("y", "<f8")]))
    v = float(case["value"])
    src["x"], src["y"] = v, -v
    dst = np.zeros(1, np.dtype([("x_head", "<f4"), ("x_tail", "<f4"), ("y_head", "<f4"), ("y_tail", "<f4")]))
    lib.fsh_convert_orbit_f64_to_p2x32(src.ctypes.data, 1, dst.ctypes.data)
    head, tail = float(dst["x_head"][0]), float(dst["x_tail"][0])
    assert (float(dst["y_head"][0]), float(dst["y_tail"][0])) == (-head, -tail)  # the split is odd
    if "sum_tolerance" in case:
        assert abs((head + tail) - v) <= case["sum_tolerance"]
    if case.get("both_words_zero"):
        assert head == 0.0 and tail == 0.0
    if case.get("closer_than_the_head_alone"):
        head_err = abs(v - float(np.float32(v)))
        assert abs(v - (head + tail)) < head_err or head_err < 1e-15
    assert abs(tail) <= abs(head) * 2.0 ** -23 or head == 0.0  # normalised: the tail is below the head's last place


@pytest.mark.parametrize("case", VEC["square_aspect_ratio"], ids=[c["name"] for c in VEC["square_aspect_ratio"]])
def test_square_aspect_ratio(native_libs, case):
    """TestPointZoomBBConverter.cpp: a view's box after PointZoomBBConverter::SquareAspectRatio(width, height) -- what inputs.View
    applies to the box it is given, as Fractal does before it renders (every pixel's delta-c is a fraction of this box)."""
    v = inputs.View(*case["box"], case["screen"][0], case["screen"][1], num_iterations=100)
    x0, y0, x1, y1 = (float(t) for t in v.bbox())
    assert abs((x1 - x0) - case["width"]) <= case["tolerance"]
    assert abs((y1 - y0) - case["height"]) <= case["tolerance"]
    # centred on the original box
    assert abs((x0 + x1) / 2) <= case["tolerance"] and abs((y0 + y1) / 2) <= case["tolerance"]
