"""GPU: bench.py end to end at a small frame -- the one-rank run, the RCCL path with a single rank (FS_FORCE_DIST=1: process
group, gather, row order, pipelined delivery) and, where the box has two GPUs, two torchrun ranks over nccl.  Every run must
print exactly one JSON line whose frame checksum equals the plain one-GPU run's and whose oracle rows are bit-exact."""
import json
import os
import subprocess
import sys

import pytest

from fractalshark_amd import GPURenderer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "960", "--height", "544", "--steps", "3", "--warmup", "1", "--no-secondary", "--cpu-sample-rows", "4"]


def _run(extra_args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FS_NO_BUILD"] = "1"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, *extra_args], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single(native_libs):
    return _run([])


def test_one_rank_line(single):
    d = single
    assert d["n_gpus"] == 1 and d["steps"] == 3
    assert d["cpu_sample_rows_bit_exact"] is True
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    ft = d["frame_timing"]
    assert ft["tile_order"].startswith("longest first")
    assert ft["latency_ms_cold"] is not None and ft["latency_ms_warm"] > 0
    # sustained frames overlap the copy-out of the previous frame: never slower than one frame at a time (10 % slack for
    # a 3-frame sample)
    assert ft["sustained_ms_per_frame"] <= ft["latency_ms_warm"] * 1.1


def test_forced_process_group_one_rank(single):
    d = _run([], {"FS_FORCE_DIST": "1", "MASTER_PORT": "29547"})
    assert d["frame_checksum"] == single["frame_checksum"]
    assert d["cpu_sample_rows_bit_exact"] is True  # the N > 1 line carries its own parity verdict
    assert d["cpu_baseline"] is None               # ... and no CPU baseline (N = 1 only)
    assert d["roofline"]["pixel_steps_per_launch"] == single["roofline"]["pixel_steps_per_launch"]


def test_natural_tile_order_flag(single):
    d = _run(["--natural-tile-order", "--no-cpu"])
    assert d["frame_checksum"] == single["frame_checksum"]
    assert d["frame_timing"]["tile_order"] == "natural" and d["frame_timing"]["latency_ms_cold"] is None


def test_two_ranks_over_nccl(single):
    n = GPURenderer.device_count()
    if n < 2:
        pytest.skip("needs >= 2 HIP devices (this box has %d)" % n)
    d = _run(["--gpus", "2"])
    assert d["n_gpus"] == 2
    assert d["frame_checksum"] == single["frame_checksum"]
    assert d["cpu_sample_rows_bit_exact"] is True
    assert d["roofline"]["pixel_steps_per_launch"] == single["roofline"]["pixel_steps_per_launch"]


@pytest.mark.parametrize("ranks", [2, 3])
def test_rank_processes_sharing_the_device_run_the_whole_of_main(single, ranks):
    """Real rank PROCESSES through the whole of bench.py's main() on a one-GPU box: `--gpus N --dist-backend gloo --share-device`
    starts N torchrun ranks that all use device 0 and exchange their slices through host memory (RCCL refuses two ranks on one
    device) -- the barriers, the max over ranks, the all-reduced step counts, rank-0-only buffers and the row-band arithmetic
    all run with a peer.  The line must carry the one-GPU frame (checksum), the oracle's rows and the summed step counts."""
    d = _run(["--gpus", str(ranks), "--dist-backend", "gloo", "--share-device"], timeout=1500)
    assert d["n_gpus"] == ranks
    assert d["config"]["exchange"].startswith("gloo")
    assert d["frame_checksum"] == single["frame_checksum"]
    assert d["cpu_sample_rows_bit_exact"] is True
    assert d["cpu_baseline"] is None
    assert d["roofline"]["pixel_steps_per_launch"] == single["roofline"]["pixel_steps_per_launch"]
    assert d["roofline"]["at_iterations_per_launch"] == single["roofline"]["at_iterations_per_launch"]


@pytest.mark.parametrize("ranks", [2, 3])
def test_direct_host_path_rank_processes_fill_one_shared_frame(single, ranks):
    """--host-path direct (round 6): no gather -- every rank process copies its own bands (fs_copy_bands_to_host) into ONE frame in
    POSIX shared memory that all ranks have page-locked, rank 0 reads the frame when every rank's counter says its bands have
    landed.  Same frame (checksum and CRC-32) as the one-GPU run, oracle rows bit-exact, summed step counts."""
    d = _run(["--gpus", str(ranks), "--dist-backend", "gloo", "--share-device", "--host-path", "direct"], timeout=1500)
    assert d["n_gpus"] == ranks
    assert d["config"]["host_path"] == "direct" and d["config"]["exchange"].startswith("none on the data path")
    assert d["frame_checksum"] == single["frame_checksum"]
    assert d["frame_crc32"] == single["frame_crc32"]
    assert d["cpu_sample_rows_bit_exact"] is True
    assert d["roofline"]["pixel_steps_per_launch"] == single["roofline"]["pixel_steps_per_launch"]


def test_direct_host_path_forced_process_group_one_rank(single):
    d = _run(["--host-path", "direct"], {"FS_FORCE_DIST": "1", "MASTER_PORT": "29549"})
    assert d["frame_crc32"] == single["frame_crc32"] and d["cpu_sample_rows_bit_exact"] is True


def test_c1_direct_line(native_libs):
    """BASELINE config C1 (View 0 1024x768, CalcCpuHDR<uint32_t,double,double>) has a bench line: the whole frame == the oracle's
    frame rendered in the same run and == the committed oracle CRC."""
    env = dict(os.environ)
    env["FS_NO_BUILD"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1_direct", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.strip()][-1])
    assert d["config"]["workload"] == "view0_1024x768_f64_direct" and d["dtype"] == "f64"
    assert d["cpu_sample_rows_bit_exact"] is True and d["frame_crc32_equals_oracle_frame"] is True
    assert d["roofline"]["kernel"] == "k_direct_f64" and d["roofline"]["frac"] > 0
    assert d["cpu_baseline"]["value"] > 0
