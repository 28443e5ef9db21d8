"""GPU: k_lav2_hdr64 (csrc/kernels_hdr64.hip, the production HDRFloat<double> LAv2 kernel of round 6: one vote per
HDRFloatComplex add, ldexp for the exact power-of-two scalings, value comparisons for the norm tests) against the literal kernel
k_lav2_lit<double> (FS_VARIANT_LITERAL) over every built-in view whose orbit stays manageable and generated views at widths
1e-8 .. 1e-31 -- all three LAv2 modes, both stage-test directions -- and against the CPU oracle (Cpu64PerturbedBLAV2HDR's
restatement).  Small frames and capped iteration counts: the point is the variety of exponent gaps the adds see (both directions,
gaps beyond 120, zero parts, rebases at the orbit's zero), not the size."""
from decimal import Decimal, getcontext

import numpy as np
import pytest

import _oracle
from fractalshark_amd import (GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_HDR64, inputs)

pytestmark = pytest.mark.gpu
W, H, CAP = 64, 40, 60000
BIG_ORBITS = {10, 15, 22}


@pytest.fixture(scope="module")
def renderer(native_libs):
    assert GPURenderer.TestCudaIsWorking() != 0, "no usable HIP device: the product path has no CPU fallback"
    r = GPURenderer(0)
    yield r
    r.set_kernel_variant(0)
    r.close()


def _pairs(co):
    return [(float(c["m"]), int(c["e"])) for c in co]


def _views():
    out = [("view%d" % n, n, None) for n in sorted(inputs.builtin_views()) if n not in BIG_ORBITS]
    centres = [("-0.5482057480704757084582125675467330293766992786373239", "-0.5775708389036038428051089822018505586755517268027721"),
               ("-1.7685736563152709932817429153295447129341", "0.0"),
               ("-0.1528465308235274786391493323577", "1.0397032701234428320367513768879")]
    for ci, c in enumerate(centres):
        for wd in ("1e-8", "1e-14", "1e-22", "1e-31"):
            out.append(("gen%d_%s" % (ci, wd), None, (c, wd)))
    return out


def _render(r, co, n, mode, parity):
    assert r.ClearMemory() == 0
    assert r.RenderPerturbLAv2(None, None, None, *co, n, T=T_HDR64, Mode=mode, parity=parity) == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n, out) == 0
    assert r.SyncComputeStream() == 0
    return out[:H, :W].copy()


@pytest.mark.parametrize("name,builtin,gen", _views(), ids=[v[0] for v in _views()])
def test_hdr64_kernel_equals_literal_kernel_and_oracle(renderer, native_libs, name, builtin, gen):
    if builtin is not None:
        v = inputs.View.builtin(builtin, W, H, antialiasing=1)
    else:
        getcontext().prec = 80
        (cx, cy), wd = gen
        cxd, cyd, w = Decimal(cx), Decimal(cy), Decimal(wd)
        h = w * H / W
        v = inputs.View(str(cxd - w / 2), str(cyd - h / 2), str(cxd + w / 2), str(cyd + h / 2), W, H, num_iterations=50000)
    ob = inputs.Orbit(v, is64=True)
    if ob.count > 2_000_000:
        pytest.skip("orbit of %d entries" % ob.count)
    la = inputs.LATable(ob)
    n = min(v.num_iterations, CAP)
    co = _pairs(v.coords_perturb(ob))
    r = renderer
    assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    try:
        for mode in (LAV2_FULL, LAV2_PO, LAV2_LAO):
            for parity in (PARITY_CPU_GPUSTAGE, PARITY_CPU):
                if mode == LAV2_PO and parity == PARITY_CPU:
                    continue  # (served by the scalar kernel: not this kernel's path)
                assert r.set_kernel_variant(0) == 0
                fast = _render(r, co, n, mode, parity)
                assert r.set_kernel_variant(1) == 0
                lit = _render(r, co, n, mode, parity)
                assert np.array_equal(fast, lit), (name, mode, parity, int((fast != lit).sum()))
                if mode == LAV2_FULL:
                    ref = _oracle.lav2_hdr32(v, ob, la, stage_test=0 if parity == PARITY_CPU else 1, n_iterations=n)
                    assert np.array_equal(fast, ref[:H, :W]), (name, "oracle", parity)
    finally:
        r.set_kernel_variant(0)


def test_hdr64_kernel_counts_equal_the_literal_kernels(renderer, native_libs):
    """The counting instantiation: executed AT iterations, LA steps and perturbation steps equal the literal kernel's (same states,
    same decisions), on a frame that uses all three phases."""
    v = inputs.View.builtin(14, 256, 144, antialiasing=1)
    ob = inputs.Orbit(v, is64=True)
    la = inputs.LATable(ob)
    co = _pairs(v.coords_perturb(ob))
    r = renderer
    assert r.InitializeMemory(256, 144, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    got = []
    try:
        for variant in (0, 1):
            assert r.set_kernel_variant(variant) == 0
            r.enable_step_count(True)
            assert r.RenderPerturbLAv2(None, None, None, *co, v.num_iterations, T=T_HDR64, Mode=LAV2_FULL,
                                       parity=PARITY_CPU_GPUSTAGE) == 0
            assert r.SyncComputeStream() == 0
            st = r.read_step_count()
            got.append((st["at_iterations"], st["la_steps"], st["perturb_steps"], st["pixels"]))
            r.enable_step_count(False)
    finally:
        r.set_kernel_variant(0)
    assert got[0] == got[1] and got[0][1] > 0 and got[0][2] > 0
