"""CPU-only checks of the 2x32 ("float-float + exponent", HDRFloat<CudaDblflt>) chain.

The reference has no CPU implementation of this type, so oracle/gpu_ref_2x32.cpp is "parity
unpinned".  What can be checked without the CUDA reference is checked here:
  * the double-float primitives against exact rational arithmetic (error bounds the reference's source quotes,
    dblflt.cuh:108-115,159-163);
  * the double -> (head, tail) conversion of the host builders (exact for 48-bit values, normalised);
  * the rendered iteration counts against the *pinned* HDRFloat<double> oracle on the same view: identical AT / LA
    work, and per-pixel counts that differ only by the GPU kernel's earlier bailout (|z|^2 >= 4 instead of > 256).
"""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

import _oracle
from fractalshark_amd import inputs


def _df(a):
    return (C.c_float * 2)(float(a[0]), float(a[1]))


def _norm_pair(rng, scale=1.0):
    """A normalised float-float: head = RN(v), tail = RN(v - head) for a random 53-bit v."""
    v = float(rng.uniform(-2, 2)) * scale
    h = np.float32(v)
    t = np.float32(v - float(h))
    return (float(h), float(t))


def _exact(p):
    return Fraction(p[0]) + Fraction(p[1])


@pytest.mark.parametrize("op,name", [(lambda x, y: x + y, "orc_df_add"), (lambda x, y: x - y, "orc_df_sub"),
                                     (lambda x, y: x * y, "orc_df_mul")])
def test_double_float_primitives_against_exact_arithmetic(op, name):
    lib = _oracle.lib()
    rng = np.random.default_rng(1234)
    worst = Fraction(0)
    for _ in range(4000):
        a = _norm_pair(rng, 2.0 ** int(rng.integers(-20, 20)))
        b = _norm_pair(rng, 2.0 ** int(rng.integers(-20, 20)))
        out = (C.c_float * 2)()
        getattr(lib, name)(_df(a), _df(b), out)
        exact = op(_exact(a), _exact(b))
        got = Fraction(out[0]) + Fraction(out[1])
        # result is normalised: head == RN(head + tail)
        assert np.float32(float(out[0]) + float(out[1])) == np.float32(out[0])
        if exact != 0:
            # add/sub are accurate relative to the larger operand (cancellation loses relative accuracy by design)
            ref = max(abs(_exact(a)), abs(_exact(b))) if name != "orc_df_mul" else abs(exact)
            worst = max(worst, abs(got - exact) / ref)
    assert worst < Fraction(1, 2 ** 43), float(worst)  # ~2^-44..-45 observed; the source quotes 2^-104 for dbldbl


def test_reduce_renormalises_head_and_shifts_tail():
    lib = _oracle.lib()
    rec = np.zeros(1, inputs.REAL_2X32)
    rec[0] = (0.7162003, 1.8376134e-10, -152)
    before = Fraction(float(rec["head"][0])) + Fraction(float(rec["tail"][0]))
    lib.orc_h2_reduce(rec.ctypes.data)
    assert 1.0 <= abs(rec["head"][0]) < 2.0 and rec["e"][0] == -153
    after = Fraction(float(rec["head"][0])) + Fraction(float(rec["tail"][0]))
    assert after == before * 2
    # zero stays zero with its exponent (HDRFloat.h:459-465)
    rec[0] = (0.0, 0.0, -7)
    lib.orc_h2_reduce(rec.ctypes.data)
    assert tuple(rec[0]) == (0.0, 0.0, -7)
    # the reference's quirk: a zero tail under a head < 1 picks up the exponent field (HDRFloat.h:474-477)
    rec[0] = (0.5, 0.0, 0)
    lib.orc_h2_reduce(rec.ctypes.data)
    assert rec["head"][0] == 1.0 and rec["e"][0] == -1 and rec["tail"][0] == np.float32(2.0 ** -126)


@pytest.fixture(scope="module")
def view5_2x32(native_libs):
    v = inputs.View.builtin(5, 48, 32, antialiasing=1)
    o = inputs.Orbit(v, is64=True)
    la = inputs.LATable(o, host_threads=8, use_small_exponents=True)
    return v, o, la, inputs.Orbit2x32(o), inputs.LATable2x32(la)


def test_conversion_from_hdr64_is_exact_to_48_bits_and_normalised(view5_2x32):
    v, o, la, o2, la2 = view5_2x32
    e64, e2 = o.entries(), o2.entries()
    assert (e64["ex"] == e2["ex"]).all() and (e64["ey"] == e2["ey"]).all()
    hx, tx = e2["x_head"].astype(np.float64), e2["x_tail"].astype(np.float64)
    # head + tail reproduces the double mantissa to 2^-48 relative; |tail| <= ulp(head)/2
    err = np.abs(hx + tx - e64["mx"])
    assert (err <= np.abs(e64["mx"]) * 2.0 ** -47).all()
    assert (np.float32(hx + tx) == e2["x_head"]).all()
    # LA records: 104 B, step lengths and links carried over unchanged
    r64 = la.records().view(np.uint32).reshape(la.count, 32)
    r2 = la2.records().view(np.uint32).reshape(la2.count, 26)
    assert (r64[:, 30:32] == r2[:, 24:26]).all()
    assert la2.at.StepLength == la.at.StepLength
    assert la2.at.ThresholdC.e == la.at.ThresholdC.e
    assert abs(float(la2.at.ThresholdC.head) + float(la2.at.ThresholdC.tail) - la.at.ThresholdC.m) < 2.0 ** -46


def test_2x32_render_tracks_the_pinned_hdr64_oracle(view5_2x32):
    v, o, la, o2, la2 = view5_2x32
    r2, st2 = _oracle.gpu_lav2_2x32(v, o2, la2, stats=True)
    r64, st64 = _oracle.lav2_hdr32(v, o, la, stage_test=1, stats=True)
    a = r2[: v.height, : v.width].astype(np.int64)
    b = r64[: v.height, : v.width].astype(np.int64)
    # the approximation part (AT + LA stages) takes exactly the same steps
    assert st2["at_iterations"] == st64["at_iterations"] and st2["la_steps"] == st64["la_steps"]
    # the perturbation part leaves 1-4 iterations earlier: the CUDA kernel bails out at |z|^2 >= 4
    # (compareToBothPositiveReducedTemplate, HDRFloat.h:1169-1184), the CPU function at |z|^2 > 256
    d = b - a
    assert ((d >= 0) & (d <= 6)).mean() > 0.98, np.unique(d, return_counts=True)
    # modes: LAO stops after the LA stages, PO skips them
    lao = _oracle.gpu_lav2_2x32(v, o2, la2, mode=2)[: v.height, : v.width]
    assert (lao <= a).all() and (lao > 0).any()
    po, stpo = _oracle.gpu_lav2_2x32(v, o2, None, mode=1, rows=(0, 2), stats=True)
    assert stpo["at_iterations"] == 0 and stpo["la_steps"] == 0
    # perturbation-only agrees with the approximated render except at a few chaotic pixels
    assert (np.abs(po[:2, : v.width].astype(np.int64) - a[:2]) <= 2).mean() > 0.8


def test_product_df32_host_build_matches_oracle_bitwise(native_libs):
    """fractalshark_amd/csrc/df32_math.hpp (the header the HIP kernel is built from, instantiated for the host in
    libfsinputs.so) against the oracle's independent restatement: double-float primitives, HDR add/sub with its 4-way
    exponent alignment, real and complex Reduce -- bit for bit on random and edge-case operands."""
    from fractalshark_amd import _capi
    hl = _capi.inputs_lib()
    ol = _oracle.lib()
    rng = np.random.default_rng(99)

    def pair(scale_exp):
        p = _norm_pair(rng, 2.0 ** scale_exp)
        return np.float32(p[0]), np.float32(p[1])

    for k in range(3000):
        a, b = pair(int(rng.integers(-30, 30))), pair(int(rng.integers(-30, 30)))
        for op, name in ((0, "orc_df_add"), (1, "orc_df_sub"), (2, "orc_df_mul")):
            o1, o2 = (C.c_float * 2)(), (C.c_float * 2)()
            hl.fsh_df32_op(op, _df(a), _df(b), o1)
            getattr(ol, name)(_df(a), _df(b), o2)
            assert bytes(o1) == bytes(o2), (op, a, b)
    recs = []
    for k in range(3000):
        ea, eb = int(rng.integers(-300, 300)), 0
        eb = ea + int(rng.choice([0, 1, -1, 5, -5, 119, -119, 120, -120, 121, -121, 126, -127, 200, -200]))
        a, b = pair(0), pair(0)
        recs.append(((a[0], a[1], ea), (b[0], b[1], eb)))
    recs.append(((0.0, 0.0, -268435456), (1.5, 1e-9, -3)))      # zero + x
    recs.append(((1.5, 1e-9, -3), (0.0, 0.0, -268435456)))      # x + zero
    recs.append(((1.25, 0.0, 7), (-1.25, 0.0, 7)))              # exact cancellation: exponent reset to MIN
    for ra, rb in recs:
        A = np.array([ra], inputs.REAL_2X32)
        B = np.array([rb], inputs.REAL_2X32)
        for sub in (0, 1):
            o1, o2 = np.zeros(1, inputs.REAL_2X32), np.zeros(1, inputs.REAL_2X32)
            hl.fsh_hr2_add(A.ctypes.data, B.ctypes.data, sub, o1.ctypes.data)
            ol.orc_h2_add(A.ctypes.data, B.ctypes.data, sub, o2.ctypes.data)
            assert o1.tobytes() == o2.tobytes(), (ra, rb, sub)
        r1, r2 = A.copy(), A.copy()
        hl.fsh_hr2_reduce(r1.ctypes.data)
        ol.orc_h2_reduce(r2.ctypes.data)
        assert r1.tobytes() == r2.tobytes()
    cdt = np.dtype([("re_head", "<f4"), ("re_tail", "<f4"), ("im_head", "<f4"), ("im_tail", "<f4"), ("e", "<i4")])
    for k in range(2000):
        re, im = pair(int(rng.integers(-40, 40))), pair(int(rng.integers(-40, 40)))
        if k % 50 == 0:
            im = (np.float32(0.0), np.float32(0.0))
        c1 = np.array([(re[0], re[1], im[0], im[1], int(rng.integers(-500, 500)))], cdt)
        c2 = c1.copy()
        hl.fsh_hc2_reduce(c1.ctypes.data)
        ol.orc_c2_reduce(c2.ctypes.data)
        assert c1.tobytes() == c2.tobytes()
