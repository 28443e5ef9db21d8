#!/usr/bin/env python3
"""Generate the small committed golden fixtures in tests/golden/ with the CPU oracle.

The oracle itself is pinned against the reference's own golden CRC-64s
(FractalSharkTest/TestRenderGoldens.cpp:84-97; tests/test_oracle_pins.py).  These fixtures are inputs and
expected iteration buffers only -- data, no reference source.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import _oracle  # noqa: E402
from fractalshark_amd import _build, inputs  # noqa: E402


def main():
    _build.build_inputs()
    out = {}
    # view 0, direct double, 64x48
    v0 = inputs.View.builtin(0, 64, 48)
    out["view0_direct_f64_64x48"] = _oracle.direct_f64(v0)
    out["view0_direct_f64_64x48_coords"] = v0.coords_direct_f64()
    # view 5, 64x36 (16:9 like the BASELINE configs)
    v5 = inputs.View.builtin(5, 64, 36)
    ob = inputs.Orbit(v5)
    la = inputs.LATable(ob)
    bla = inputs.BLATable(ob)
    out["view5_orbit_count_period"] = np.array([ob.count, ob.period], np.uint64)
    out["view5_orbit_head"] = ob.entries()[:64].copy()
    out["view5_orbit_tail"] = ob.entries()[-64:].copy()
    out["view5_la_count_stages"] = np.array([la.count, la.stage_count, int(la.use_at)], np.uint32)
    out["view5_la_stages"] = la.stages().copy()
    out["view5_la_head"] = la.records()[:32].copy()
    out["view5_coords_hdr32_64x36"] = v5.coords_perturb_hdr32(ob)
    out["view5_lav2_cpu_64x36"] = _oracle.lav2_hdr32(v5, ob, la, stage_test=0)
    out["view5_lav2_gpustage_64x36"] = _oracle.lav2_hdr32(v5, ob, la, stage_test=1)
    out["view5_lao_cpu_64x36"] = _oracle.lav2_hdr32(v5, ob, la, stage_test=0, mode=2)
    out["view5_po_64x36"] = _oracle.bla_hdr32(v5, ob, None)
    out["view5_bla_64x36"] = _oracle.bla_hdr32(v5, ob, bla)
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
