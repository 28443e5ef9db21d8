#!/usr/bin/env python3
"""Generate tests/golden/golden_crc.json: for each of the reference's twelve golden render cases
(FractalSharkTest/TestRenderGoldens.cpp:84-97) the CRC-32 of the 256x256 iteration buffer the CPU oracle produces --
written ONLY when the oracle's PNG reproduces the reference's CRC-64 literal for that case (needs oracle/_ref, i.e.
/root/reference at build time).  The GPU tests then hold the HIP output against both: the PNG CRC-64 literal (through
oracle/_ref/libpngpin.so when it travelled to the GPU box) and this buffer CRC (always).

Run:  python tests/golden/make_golden_crc.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import _oracle  # noqa: E402
import golden_cases as gc  # noqa: E402
from fractalshark_amd import _build, inputs  # noqa: E402


def main():
    _build.build_inputs()
    assert _oracle.pin_lib() is not None, "oracle/_ref/libpngpin.so is needed (make -C oracle _ref)"
    out = {}
    for name, view_n, alg, aa, crc64 in gc.CASES:
        v, ob, table = gc.build_inputs(inputs, view_n, alg, aa)
        it = gc.oracle_render(_oracle, alg, v, ob, table, aa)
        got = _oracle.png_crc64(it, gc.W, gc.H, aa, v.num_iterations)
        assert got == crc64, (name, got, crc64)
        out[name] = {"algorithm": alg, "view": view_n, "antialiasing": aa, "png_crc64": crc64,
                     "iter_buffer_crc32": gc.buffer_crc32(it), "iter_buffer_shape": list(it.shape),
                     "iter_sum": int(it.astype("uint64").sum())}
        print(name, out[name])
    with open(os.path.join(HERE, "golden_crc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
