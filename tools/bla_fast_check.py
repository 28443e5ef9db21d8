"""A/B of the hand-written BLA kernel (default) against the compiled one (variant 2) and, at small sizes, the CPU oracle.
Usage: python tools/bla_fast_check.py [view width height [oracle]] ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fractalshark_amd import GPURenderer, T_HDR32, inputs  # noqa: E402


def run(view, w, h, oracle):
    v = inputs.View.builtin(view, w, h, antialiasing=1)
    o = inputs.Orbit(v)
    bla = inputs.BLATable(o)
    co = v.coords_perturb(o)
    r = GPURenderer(0)
    assert r.InitializeMemory(w, h, 1, None, 0, 0, 0, False) == 0
    lib = r._lib
    assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, o.data_ptr, o.count, o.count, o.period) == 0
    assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
    out = {}
    for name, variant in (("compiled", 2), ("asm", 0)):
        assert r.set_kernel_variant(variant) == 0
        assert r.ClearMemory() == 0
        ms = []
        for _ in range(2):
            assert lib.fs_render_bla(r._h, T_HDR32, co.ctypes.data, v.num_iterations) == 0
            assert r.SyncComputeStream() == 0
            ms.append(r.last_kernel_ms())
        buf = r.new_iter_buffer()
        assert r.RenderCurrent(v.num_iterations, buf) == 0
        assert r.SyncComputeStream() == 0
        out[name] = (buf, min(ms))
    if os.environ.get("FS_BLA_FAST_PROBE") == "1":
        import ctypes as C
        raw = (C.c_uint64 * 40)()
        assert lib.fs_read_stats_raw(r._h, raw, 40) == 0
        e, ss, sl, wv = list(raw)[20:24]
        print("  step passes with every lane at ONE orbit entry: %.3f of %d; jump passes with every lane at ONE record: %.3f of %d"
              % (raw[31] / max(1, raw[28]), raw[28], raw[32] / max(1, raw[27]), raw[27]), flush=True)
        print("  passes per wave: " + ", ".join("%s %.1f" % (k, x / max(1, wv)) for k, x in zip(
            ["lookup", "pre-test", "ladder round", "jump", "step", "step with z", "rebase"], list(raw)[24:31])), flush=True)
        print("  probe (all asm launches): statement entered %.1f times per wave, literal step %.2f, literal lookup round %.2f"
              % (e / max(1, wv), ss / max(1, wv), sl / max(1, wv)), flush=True)
    same = np.array_equal(out["compiled"][0], out["asm"][0])
    msg = "view %d %dx%d orbit %d: compiled %.3f ms, asm %.3f ms, identical %s" % (view, w, h, o.count, out["compiled"][1],
                                                                                  out["asm"][1], same)
    if not same:
        d = out["compiled"][0] != out["asm"][0]
        ys, xs = np.nonzero(d)
        msg += " (%d pixels differ, first at x=%d y=%d: %d vs %d)" % (d.sum(), xs[0], ys[0], out["compiled"][0][ys[0], xs[0]],
                                                                      out["asm"][0][ys[0], xs[0]])
    if oracle:
        import _oracle
        ref = _oracle.bla_hdr32(v, o, bla)
        msg += "; == oracle %s" % np.array_equal(ref, out["asm"][0])
    print(msg, flush=True)
    r.close()
    return same


if __name__ == "__main__":
    args = sys.argv[1:]
    cases = []
    while args:
        view, w, h = int(args[0]), int(args[1]), int(args[2])
        args = args[3:]
        orc = bool(args) and args[0] == "oracle"
        if orc:
            args = args[1:]
        cases.append((view, w, h, orc))
    if not cases:
        cases = [(5, 64, 36, True), (19, 64, 36, True), (19, 640, 360, False), (19, 3840, 2160, False)]
    ok = all([run(*c) for c in cases])
    sys.exit(0 if ok else 1)
