"""The hand-issued orbit loads of the tuned loops (asm s_load_dwordx4 / global_load_dwordx3 with separate s_waitcnt
statements) are outside the compiler's wait-count bookkeeping.  tools/check_inflight_loads.py disassembles the built gfx950
code objects and proves, by a may-analysis over every function's control-flow graph, that no instruction names a
register a load may still be writing.  Runs on the CPU: it inspects the library, it does not execute it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_inflight_loads as chk  # noqa: E402


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_instruction_touches_a_register_of_a_load_in_flight(native_libs):
    from fractalshark_amd import _build
    funcs, findings, n_scalar, n_vector = chk.run(_build.LIB_RENDER)
    tuned = [n for n in funcs if "k_lav2_hdr32_fast" in n or "k_perturb_scalar" in n]
    assert len(tuned) >= 8 and n_scalar > 500 and n_vector > 300
    assert not findings, findings[:5]


def test_the_analysis_sees_a_planted_hazard():
    mk = lambda a, t: {"addr": a, "size": 4, "op": t.split(None, 1)[0], "ops": t.split(None, 1)[1] if " " in t else "",
                       "text": t}
    ok = [mk(0, "s_load_dwordx4 s[8:11], s[0:1], 0x0"), mk(8, "v_mov_b32_e32 v1, v2"), mk(12, "s_waitcnt lgkmcnt(0)"),
          mk(16, "v_mov_b32_e32 v3, s9"), mk(20, "s_endpgm")]
    assert chk.check_function("ok", ok) == []
    bad = [mk(0, "s_load_dwordx4 s[8:11], s[0:1], 0x0"), mk(8, "s_mov_b32 s20, s9"), mk(12, "s_waitcnt lgkmcnt(0)"),
           mk(16, "s_endpgm")]
    assert len(chk.check_function("bad", bad)) == 1
    # in-order vector loads: vmcnt(1) retires all but the youngest
    vec = [mk(0, "global_load_dwordx3 v[4:6], v7, s[12:13]"), mk(8, "global_load_dwordx3 v[8:10], v7, s[12:13] offset:16"),
           mk(16, "s_waitcnt vmcnt(1)"), mk(20, "v_add_f32_e32 v1, v4, v5"), mk(24, "v_add_f32_e32 v1, v8, v9"),
           mk(28, "s_endpgm")]
    got = chk.check_function("vec", vec)
    assert len(got) == 1 and got[0][1].startswith("v_add_f32_e32 v1, v8")
    # a branch around the wait: the join still has the load in flight
    br = [mk(0, "s_load_dword s4, s[0:1], 0x0"), mk(8, "s_cbranch_scc1 1"), mk(12, "s_waitcnt lgkmcnt(0)"),
          mk(16, "s_add_u32 s5, s4, 1"), mk(20, "s_endpgm")]
    assert len(chk.check_function("br", br)) == 1
