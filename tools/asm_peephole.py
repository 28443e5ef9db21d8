#!/usr/bin/env python3
"""Peephole over the compiler's gfx950 assembly (an A/B build step of fractalshark_amd/_build.py: FS_PEEPHOLE_UNITS; NOT part of the
product build -- measured on k_lav2_hdr64, same box: 33.33 ms with 838 selects rewritten against 33.28 without).

    v_cndmask_b32_e32 vD, src0, vS1, vcc      ->      v_cndmask_b32_e64 vD, src0, vS1, vcc

Measured on MI355X (tools/microbench/valu_rates_f64.hip, profiles/r06_valu_issue_rates_f64.jsonl): the 32-bit VOP2 encoding of
v_cndmask_b32 issues one wave64 instruction per SIMD every 23.4 cycles with 8 waves per SIMD, the 64-bit VOP3 encoding of the SAME
operation with the SAME mask register (vcc) every 4.4 -- like every other VOP3 instruction.  The compiler always shrinks a select
whose mask lives in vcc to the 32-bit form (SIShrinkInstructions; no switch turns that off), so the compiled kernels pay five
instruction slots for every select.  The rewrite changes the encoding only: same operands, same result.  Left alone: a select whose
src0 is a 32-bit literal (VOP3 has no literals on gfx9) or a scalar register (vcc + an SGPR would be two constant-bus reads).

  python tools/asm_peephole.py in.s out.s      (prints the number of rewritten instructions)"""
import re
import sys

PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+),\s*([^,]+?),\s*(v\d+),\s*vcc\s*$")
INLINE_F = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0"}


def src0_ok(tok):
    tok = tok.strip()
    if re.fullmatch(r"v\d+", tok):
        return True
    if tok in INLINE_F:
        return True
    if re.fullmatch(r"-?\d+", tok):
        return -16 <= int(tok) <= 64
    return False  # literal (0x..., large numbers), sgpr, special registers


def rewrite(text):
    out, n = [], 0
    for ln in text.split("\n"):
        body = ln.split(";", 1)[0].rstrip()
        m = PAT.match(body)
        if m and src0_ok(m.group(4)):
            out.append("%sv_cndmask_b32_e64%s%s, %s, %s, vcc" % (m.group(1), m.group(2), m.group(3), m.group(4).strip(), m.group(5)))
            n += 1
        else:
            out.append(ln)
    return "\n".join(out), n


if __name__ == "__main__":
    src = open(sys.argv[1]).read()
    dst, n = rewrite(src)
    open(sys.argv[2], "w").write(dst)
    print("%s: %d v_cndmask_b32_e32 -> e64, %d left" % (sys.argv[1], n, len(re.findall(r"v_cndmask_b32_e32", dst))))
