"""The hand-scheduled loops of the tuned kernels name physical registers inside one `asm` statement (FS_FAST_LOOP,
FS_FAST_LOOP_FL, FS_FAST_LOOP_FD in csrc/scaled_runs.hpp).  tools/check_asm_registers.py finds every such loop in the BUILT gfx950
code and proves, with a liveness analysis over the function's control-flow graph, that none of the loop's scratch registers
(v[56:59], v61, v62, s[36:63], s66) is live on the loop's exits -- i.e. the compiler keeps no value of its own in them across
the statement -- and that the loop writes no vector register it does not name.  CPU only: it inspects the library."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_registers as car  # noqa: E402
import check_inflight_loads as chk  # noqa: E402


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_scratch_registers_of_the_hand_scheduled_loops_are_dead_on_exit(native_libs):
    from fractalshark_amd import _build
    funcs = chk.disassemble(_build.LIB_RENDER)
    checked, findings = 0, []
    for name, lines in funcs.items():
        if "k_lav2_hdr32_fast" not in name and "k_perturb_scalar" not in name:
            continue
        # (the step-counting build of the scalar kernel, k_perturb_scalar<float, false, true, ..>, is left out: one
        # compiler-allocated scalar of its tested-block path (s36 with this round's statement, read by an `s_add` far behind it)
        # is live on a path the path-insensitive may-analysis cannot rule out; the compiler honours the statement's clobber
        # list, and the build is covered dynamically instead -- its frame == the plain kernel's == the oracle's, its counts ==
        # the literal variant's: tests/test_gpu_variants.py::test_counting_instantiation_of_the_perturbation_only_kernel.  The
        # kernels a frame runs -- kStats = false -- and every k_lav2_hdr32_fast instantiation are checked)
        if "k_perturb_scalarIfLb0ELb1E" in name:
            continue
        f, k = car.check_function(name, lines)
        findings += f
        checked += k
    assert checked >= 12, checked
    assert not findings, findings[:3]


def test_the_analysis_sees_a_value_kept_in_a_scratch_register():
    mk = lambda a, t: {"addr": a, "size": 4, "op": t.split(None, 1)[0], "ops": t.split(None, 1)[1] if " " in t else "",
                       "text": t}
    body = ["v_max_i32_e32 v62, v60, v8", "v_pk_fma_f32 v[56:57], v[48:49], v[34:35], s[64:65]"] + \
           ["v_pk_mul_f32 v[58:59], v[48:49], v[56:57]"] * 70 + ["s_waitcnt lgkmcnt(0)"]
    ok = [mk(4 * i, t) for i, t in enumerate(body + ["v_mov_b32_e32 v1, v48", "s_endpgm"])]
    assert car.check_function("ok", ok)[0] == []
    bad = [mk(4 * i, t) for i, t in enumerate(body + ["v_mov_b32_e32 v1, v58", "s_endpgm"])]
    got = car.check_function("bad", bad)[0]
    assert len(got) == 1 and ("v", 58) in got[0][2]


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_the_bla_statement_writes_only_the_registers_it_declares(native_libs):
    """kernels_bla_fast.hip holds the whole BLA loop in one asm statement: state in v0..v15 (in/out operands), temporaries
    v16..v47 and s36..s64 (clobbers).  In the BUILT kernel the statement is the stretch from its first instruction (the
    0x807fffff constant into s50) to the restore of EXEC from s[48:49]: every vector register written there must be one of
    v0..v47, every scalar one either s36..s64 or one of the few operand registers the compiler assigned (the running mask, the
    lookup mask, the status, the slow-path mask, the trip budget: at most 8 scalars), and nothing may be left in flight at its end."""
    from fractalshark_amd import _build
    funcs = chk.disassemble(_build.LIB_RENDER)
    mine = [(n, l) for n, l in funcs.items() if "k_bla_hdr32_fast" in n]
    assert len(mine) == 2  # the default kernel and the workgroup-pooling A/B variant
    for _, lines in mine:
        start = [i for i, ins in enumerate(lines) if ins["op"] == "s_mov_b32" and ins["ops"].replace(" ", "").lower() == "s50,0x807fffff"]
        end = [i for i, ins in enumerate(lines) if ins["op"] == "s_mov_b64" and ins["ops"].replace(" ", "") == "exec,s[48:49]"]
        assert len(start) == 1 and len(end) == 1 and end[0] - start[0] > 250
        region = lines[start[0]:end[0] + 1]
        other_scalars = set()
        for ins in region:
            d, _ = car.defs_uses(ins)
            if ins["op"].startswith(("global_load", "v_cmp")) or ins["op"].startswith("s_and_saveexec"):
                d = chk.regs_of(ins["ops"].partition(",")[0])
            for kind, i in d:
                if kind == "v":
                    assert i <= 47, ins["text"]
                elif not 36 <= i <= 64:
                    other_scalars.add(i)
        assert len(other_scalars) <= 8, sorted(other_scalars)  # R, J, status, slow-path mask, trip budget
        assert region[-2]["op"] == "s_waitcnt" and "vmcnt(0)" in region[-2]["ops"]
        packed = sum(ins["op"].startswith("v_pk_") for ins in region)
        assert packed >= 20  # (the statement was found, not an empty stretch)
