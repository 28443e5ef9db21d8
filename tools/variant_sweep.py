"""Robustness sweep on the GPU: the tuned LAv2 / perturbation-only kernels (with and without scaled runs) against the
literal transcription, bit for bit, over every built-in view and a set of generated ones (zoom widths 1e-8 .. 1e-40 around
two centres, on and off the real axis).  The literal kernel is itself pinned to the CPU oracle by tests/; this sweep
widens the set of orbits the fast paths have seen.  Usage: python tools/variant_sweep.py [--size 96x54] [--po-cap 60000]"""
import argparse
import os
import sys
import time
from decimal import Decimal, getcontext

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, PARITY_CPU, T_HDR32, inputs  # noqa: E402


def render(r, v, ob, la, mode, variant, n_iter):
    assert r.set_kernel_variant(variant) == 0
    assert r.InitializeMemory(v.width, v.height, 1, None, 0, 0, 0, False) == 0
    assert r.InitializePerturb(0, ob, 0, None, la) == 0
    assert r.ClearMemory() == 0
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(ob)]
    assert r.RenderPerturbLAv2(None, None, None, *co, n_iter, T=T_HDR32, Mode=mode, parity=PARITY_CPU) == 0
    assert r.SyncComputeStream() == 0
    out = r.new_iter_buffer()
    assert r.RenderCurrent(n_iter, out) == 0
    assert r.SyncComputeStream() == 0
    return out[: v.height, : v.width].copy()


def generated(W, H):
    getcontext().prec = 80
    centres = [("-0.5482057480704757084582125675467330293766992786373239", "-0.5775708389036038428051089822018505586755517268027721"),
               ("-1.7685736563152709932817429153295447129341", "0.0"),
               ("-0.1528465308235274786391493323577", "1.0397032701234428320367513768879")]
    for ci, (cx, cy) in enumerate(centres):
        for wd in ("1e-8", "1e-14", "1e-22", "1e-31", "1e-40"):
            cxd, cyd, w = Decimal(cx), Decimal(cy), Decimal(wd)
            h = w * H / W
            yield "gen%d_%s" % (ci, wd), inputs.View(str(cxd - w / 2), str(cyd - h / 2), str(cxd + w / 2), str(cyd + h / 2),
                                                    W, H, num_iterations=50000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="96x54")
    ap.add_argument("--po-cap", type=int, default=60000)
    a = ap.parse_args()
    W, H = [int(x) for x in a.size.split("x")]
    r = GPURenderer(0)
    views = [("view%d" % n, inputs.View.builtin(n, W, H, antialiasing=1)) for n in inputs.builtin_views()]
    views += list(generated(W, H))
    bad = 0
    for name, v in views:
        t0 = time.time()
        try:
            ob = inputs.Orbit(v)
            la = inputs.LATable(ob)
        except Exception as e:  # a view the float-exponent inputs cannot express
            print(name, "skipped:", e)
            continue
        res = []
        for mode, n_iter in ((LAV2_FULL, v.num_iterations), (LAV2_PO, min(v.num_iterations, a.po_cap))):
            lit = render(r, v, ob, la, mode, 1, n_iter)
            same = [bool(np.array_equal(lit, render(r, v, ob, la, mode, k, n_iter))) for k in (0, 2)]
            res.append((same, int(lit.min()), int(lit.max())))
            bad += same.count(False)
        print("%-14s orbit %7d  full %s iters %d..%d  po %s iters %d..%d  (%.1f s)" %
              (name, ob.count, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], res[1][2], time.time() - t0), flush=True)
    r.set_kernel_variant(0)
    print("MISMATCHES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
