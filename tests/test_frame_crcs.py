"""CPU-only: tests/golden/frame_crcs.json -- whole BASELINE frames as the CPU oracle renders them (tests/golden/make_frame_crcs.py)
-- is complete, oracle-made, and reproducible: one band of the cheapest frame is rendered again by the oracle here and must
give the committed band CRC.  (The GPU side of the pin: tests/test_gpu_full_size.py asserts the HIP frame's CRC-32 and sum against
this file.)"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TABLE = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_crcs.json")))

KEYS = ["view5_3840x2160_hdrx32_lav2_full|cpu|4718592", "view5_3840x2160_hdrx32_lav2_full|cpu_gpustage|4718592",
        "view5_1920x1080_hdrx32_po|cpu|4718592", "view19_7680x4320_hdrx32_bla|cpu|113246208",
        "view14_15360x8640_hdrx64_lav2_full_aa4|cpu_gpustage|2147483646", "view14_15360x8640_hdrx2x32_lav2_full_aa4|cpu|2147483646",
        "view14_3840x2160_hdrx32_scaled_aa1_itercap|cpu|65536"]


def test_every_bench_workload_has_an_oracle_made_frame():
    import pytest
    missing = [k for k in KEYS if k not in TABLE]
    if missing and not TABLE.get("_complete", False):
        pytest.skip("make_frame_crcs.py has not rendered yet: %s" % ", ".join(missing))
    for k in KEYS:
        assert k in TABLE, k
        rec = TABLE[k]
        assert rec["source"] == "oracle"
        assert len(rec["crc32"]) == 8 and int(rec["sum"]) > 0
        nb = (rec["height"] + rec["band_rows"] - 1) // rec["band_rows"]
        assert len(rec["band_crc32"]) == nb and rec["band_rows"] % 8 == 0


def test_oracle_reproduces_a_committed_band(native_libs):
    import _oracle
    import bench
    key = "view19_7680x4320_hdrx32_bla|cpu|113246208"
    rec = TABLE[key]
    inp = bench.make_inputs("c5_bla")
    assert "%s|%s|%d" % (inp["key"], inp["parity"], inp["n_iter"]) == key
    band = 7
    y0 = band * rec["band_rows"]
    y1 = min(inp["H"], y0 + rec["band_rows"])
    buf = _oracle.workload_rows(inp, y0, y1, threads=max(1, (os.cpu_count() or 2)))
    rows = np.ascontiguousarray(buf[y0:y1, :inp["W"]]).astype("<u4", copy=False)
    assert "%08x" % (zlib.crc32(rows.tobytes()) & 0xFFFFFFFF) == rec["band_crc32"][band]
