"""GPU: the exits of k_lav2_hdr64's hand-written statements (csrc/la_step_asm.hpp, csrc/pt_step_asm.hpp) that no view reaches by
itself.  Status 2 -- a norm below 2^-1000 in a rebase / escape test, an abnormal complex0 at a rebase: the statement hands the step's
tests to the compiled code -- needs an exact zero in the middle of an orbit; tools/hdr64_statement_coverage.py counts 0 such exits over
all the parity cases.  A build with the threshold at 1e300 (-DFS_H64_ASM_TINY=1e300) takes that exit on EVERY step; its production
kernel must still render what the literal kernel renders.  The variant library is compiled here (hipcc, ~25 s) and run in a process of
its own (FSMI355_LIB is read when the library is loaded)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_step_through_status_2_renders_the_same_frames(native_libs):
    from fractalshark_amd import _build
    lib = _build.status2_test_variant()
    env = dict(os.environ, FSMI355_LIB=lib, FS_NO_BUILD="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hdr64_forced_status2_check.py")], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert p.stdout.count("identical") == 18 and "DIFFERENT" not in p.stdout, p.stdout[-3000:]
