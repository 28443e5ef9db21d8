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
