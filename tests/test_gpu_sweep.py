"""GPU: the perturbation-only / BLA kernel (k_perturb_scalar: scaled runs with the hand-scheduled loop, tuned single steps,
device-native BLA table) against the CPU oracle over every built-in view whose reference orbit stays below two million
entries and over generated views at zoom widths 1e-8 .. 1e-31 around three centres (one on the real axis) -- small frames,
capped iteration counts: the point is the variety of orbits, not the size.  tools/variant_sweep.py is the companion for the
LAv2 kernels (tuned against literal, all views)."""
from decimal import Decimal, getcontext

import numpy as np
import pytest

import _oracle
from fractalshark_amd import GPURenderer, inputs

pytestmark = pytest.mark.gpu
W, H, CAP = 48, 27, 20000
BIG_ORBITS = {10, 15, 22}  # 30 - 80 million entries: minutes of GMP each (tools/variant_sweep.py renders them)


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


@pytest.mark.parametrize("name,builtin,gen", _views(), ids=[v[0] for v in _views()])
def test_perturbation_only_and_bla_against_the_oracle(renderer, native_libs, name, builtin, gen):
    if builtin is not None:
        v = inputs.View.builtin(builtin, W, H, antialiasing=1)
    else:
        getcontext().prec = 80
        (cx, cy), wd = gen
        cxd, cyd, w = Decimal(cx), Decimal(cy), Decimal(wd)
        h = w * H / W
        v = inputs.View(str(cxd - w / 2), str(cyd - h / 2), str(cxd + w / 2), str(cyd + h / 2), W, H, num_iterations=50000)
    try:
        ob = inputs.Orbit(v)
    except Exception as e:  # a view the float-exponent inputs cannot express
        pytest.skip(str(e))
    if ob.count > 2_000_000:
        pytest.skip("orbit of %d entries" % ob.count)
    n = min(v.num_iterations, CAP)
    r = renderer
    r.set_kernel_variant(0)
    co = _pairs(v.coords_perturb(ob))
    tables = [None]
    if ob.count >= 8:
        tables.append(inputs.BLATable(ob))
    for bla in tables:
        assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
        assert r.ClearMemory() == 0
        assert r.RenderPerturbBLA(None, ob, bla, None, None, *co, n) == 0
        out = r.new_iter_buffer()
        assert r.RenderCurrent(n, out) == 0
        assert r.SyncComputeStream() == 0
        ref = _oracle.bla_hdr32(v, ob, bla, n_iterations=n)
        assert np.array_equal(out[:H, :W], ref[:H, :W]), (name, "bla" if bla is not None else "perturbation only")


@pytest.mark.parametrize("name,builtin,gen", _views(), ids=[v[0] for v in _views()])
def test_perturbation_only_long_runs_tuned_equals_literal(renderer, native_libs, name, builtin, gen):
    """The same views at a cap that gives the tuned kernel its long scaled runs (2048-step runs, 16-step bodies with one
    verdict per body and the roll-backs behind it: statuses 3 and 4 of FS_FAST_LOOP_FD16P): the tuned frame against the
    statement-for-statement variant's, which the test above ties to the oracle."""
    w, h, cap = 128, 72, 300000
    if builtin is not None:
        v = inputs.View.builtin(builtin, w, h, antialiasing=1)
    else:
        getcontext().prec = 80
        (cx, cy), wd = gen
        cxd, cyd, ww = Decimal(cx), Decimal(cy), Decimal(wd)
        hh = ww * h / w
        v = inputs.View(str(cxd - ww / 2), str(cyd - hh / 2), str(cxd + ww / 2), str(cyd + hh / 2), w, h, num_iterations=cap)
    try:
        ob = inputs.Orbit(v)
    except Exception as e:
        pytest.skip(str(e))
    if ob.count > 2_000_000:
        pytest.skip("orbit of %d entries" % ob.count)
    n = min(v.num_iterations, cap)
    r = renderer
    co = _pairs(v.coords_perturb(ob))
    frames = []
    try:
        for variant in (0, 1):
            assert r.set_kernel_variant(variant) == 0
            assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
            assert r.ClearMemory() == 0
            assert r.RenderPerturbBLA(None, ob, None, None, None, *co, n) == 0
            out = r.new_iter_buffer()
            assert r.RenderCurrent(n, out) == 0
            assert r.SyncComputeStream() == 0
            frames.append(out[:h, :w].copy())
    finally:
        r.set_kernel_variant(0)
    assert np.array_equal(frames[0], frames[1]), (name, int((frames[0] != frames[1]).sum()))
