"""GPU: the BASELINE configurations at their FULL sizes, end to end through bench.py (C-ABI renderer, pipelined frames, the copy
to the host): the WHOLE frame must equal the frame the CPU oracle renders -- CRC-32 of the valid region and the sum of the
counts against tests/golden/frame_crcs.json, which tests/golden/make_frame_crcs.py wrote in the build container from full
oracle frames (hours of CPU, no GPU involved; the reference's own pattern: TestRenderGoldens.cpp:84-97 pins whole frames by a
CRC) -- and a few rows of the same frame, spread over its height, must equal the oracle bit for bit in the same run.
C1 1024x768 (the direct binary64 kernel; the oracle's whole frame in the same run), C3 3840x2160, C2 1920x1080, C5 7680x4320, C4 15360x8640 in its three forms.  One frame buffer of C4 is 531 MB; the oracle's
share of a test run is bounded by --cpu-sample-rows."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every bench workload must have its oracle frame (set once tests/golden/make_frame_crcs.py has finished all seven)
STRICT = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_crcs.json"))).get("_complete", False)


def _bench(workload, rows, timeout=1500):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FS_NO_BUILD"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1",
           "--no-secondary", "--cpu-sample-rows", str(rows)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,rows,name", [
    ("c1_direct", 768, "view0_1024x768_f64_direct"),
    ("c3_lav2", 8, "view5_3840x2160_hdrx32_lav2_full"),
    ("c2_po", 2, "view5_1920x1080_hdrx32_po"),
    ("c5_bla", 4, "view19_7680x4320_hdrx32_bla"),
    ("c4_hdr64", 4, "view14_15360x8640_hdrx64_lav2_full_aa4"),
    ("c4_2x32", 2, "view14_15360x8640_hdrx2x32_lav2_full_aa4"),
    ("c4_scaled", 4, "view14_3840x2160_hdrx32_scaled_aa1_itercap"),
])
def test_baseline_configuration_at_full_size(native_libs, workload, rows, name):
    d = _bench(workload, rows)
    assert d["config"]["workload"] == name
    assert d["n_gpus"] == 1
    assert d["cpu_sample_rows_bit_exact"] is True, d.get("cpu_baseline")
    # the whole frame, not only the sampled rows: its CRC-32 and its sum are those of the ORACLE's frame
    if d["frame_crc32_equals_oracle_frame"] is None and not STRICT:
        pytest.skip("the oracle's frame of %s is not in tests/golden/frame_crcs.json yet (make_frame_crcs.py still rendering)" % name)
    assert d["frame_crc32_equals_oracle_frame"] is True, (d["frame_crc32"], "no oracle frame committed" if
                                                         d["frame_crc32_equals_oracle_frame"] is None else "differs")
    assert d["frame_checksum_equals_oracle_frame"] is True, d["frame_checksum"]
    assert d["roofline"]["kernel_ms"] > 0 and d["value"] > 0


@pytest.mark.parametrize("world", [2, 8])
def test_c3_full_size_row_tiled_over_a_group(native_libs, world):
    """The N > 1 data path at the BASELINE size: C3's 3840x2160 frame row-tiled over `world` members of an fs_group (they share
    this box's one device: interleaved bands rendered concurrently, gathered and re-ordered on the device), its reduction's sum
    and CRC-32 == the oracle frame's, twice (buffers and tile costs are reused by the second frame)."""
    import numpy as np
    from fractalshark_amd import GPURendererGroup, LAV2_FULL, PARITY_CPU, _capi, inputs
    import zlib
    orc = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_crcs.json"))).get("view5_3840x2160_hdrx32_lav2_full|cpu|4718592")
    if orc is None and not STRICT:
        pytest.skip("the oracle's C3 frame is not in tests/golden/frame_crcs.json yet")
    assert orc["source"] == "oracle"
    want, want_crc = int(orc["sum"]), orc["crc32"]
    v = inputs.View.builtin(5, 3840, 2160, antialiasing=1)
    ob = inputs.Orbit(v)
    la = inputs.LATable(ob)
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb_hdr32(ob)]
    g = GPURendererGroup([0] * world)
    try:
        assert g.size == world
        assert g.InitializeMemory(v.width, v.height, 1) == 0
        assert g.InitializePerturb(1, ob, la) == 0
        for _ in range(2):
            assert g.ClearMemory() == 0
            assert g.RenderPerturbLAv2(*co, v.num_iterations, Mode=LAV2_FULL, parity=PARITY_CPU) == 0
            out = g.new_iter_buffer()
            red = _capi.Reduction()
            assert g.RenderCurrent(v.num_iterations, out, red) == 0
            assert g.Sync() == 0
            assert int(out[:v.height, :v.width].astype(np.uint64).sum()) == want
            assert red.Sum == want
            assert "%08x" % (zlib.crc32(np.ascontiguousarray(out[:v.height, :v.width]).astype("<u4").tobytes()) & 0xFFFFFFFF) == want_crc
    finally:
        g.close()
