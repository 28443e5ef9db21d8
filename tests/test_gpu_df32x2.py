"""GPU: the packed double-float pair type (df32x2, csrc/df32_math.hpp) against the scalar df32 operators it mirrors --
bit for bit, per half, on a million random and special operand pairs.  The 2x32 kernels' AT loop and LA step are built
on it; their frames are checked against the oracle elsewhere, this is the unit test underneath."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_df32x2_matches_df32_per_half(tmp_path):
    exe = str(tmp_path / "df32x2_check")
    # the product is built with -ffp-contract=off (part of the numerical contract, fractalshark_amd/_build.py)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-o", exe,
           os.path.join(ROOT, "tests", "shim", "df32x2_check.hip")]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF"))}
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert p.returncode == 0, p.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["pairs"] == 1 << 19
    assert (d["add_mismatch"], d["mul_mismatch"], d["sub_mismatch"], d["swap_mismatch"]) == (0, 0, 0, 0)
    # the packed pair of extended-exponent additions of the 2x32 perturbation step: bit for bit where it claims coverage, and
    # it claims it for nearly all operand pairs (exact cancellations are the literal path's)
    assert (d["sub_lo_add_hi_mismatch"], d["hr_add2_mismatch"], d["mul_by_float_mismatch"]) == (0, 0, 0)
    assert d["hr_add2_rare"] < d["pairs"] // 4
