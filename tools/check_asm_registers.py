"""Static check of the hand-scheduled loops' register ownership (gfx950 code objects of libfsmi355.so; CPU only).

FS_FAST_LOOP_FL / FS_FAST_LOOP_FD / FS_FAST_LOOP_FD16 are single `asm` statements that NAME their registers: the state pairs
v[48:55], the temporaries v[56:59], v61, v62, the entries s[36:63] and s66, with s[64:65], s67, s[68:69] and v60 as in/out
operands.  The compiler allocates everything else around the statement and honours the clobber list -- this check makes a
break of that contract visible in the BUILT code instead of in a frame:

  * the loop is located in the disassembly by its first packed instruction (`v_pk_fma_f32 v[56:57], v[48:49], .., s[64:65]`)
    and followed to the `s_waitcnt lgkmcnt(0)` every exit ends in;
  * a backward liveness analysis over the function's control-flow graph gives the registers that are live on the region's
    exits;  the statement's pure scratch -- v[56:59], v61, v62 (declared outputs nothing reads), s[36:63], s66 (clobbers) --
    must be DEAD there: a live one would mean the surrounding code expects a value the loop has overwritten;
  * inside the region only the named registers, the operand registers the compiler assigned (at most a dozen scalars and the
    five vector inputs) and vcc / scc / exec may be written.

Usage: python tools/check_asm_registers.py [path/to/libfsmi355.so]"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_inflight_loads as chk  # noqa: E402

SCRATCH = {("v", i) for i in (56, 57, 58, 59, 61, 62)} | {("s", i) for i in list(range(36, 64)) + [66]}
# the 16-step body (FS_FAST_LOOP_FD16): entries in s[36:67] with s[66:67] an in/out operand, block bounds in s[72:75] with s75 in/out
SCRATCH16 = {("v", i) for i in (56, 57, 58, 59, 61, 62)} | {("s", i) for i in list(range(36, 66)) + [72, 73, 74]}
NAMED = {("v", i) for i in range(45, 64)} | {("s", i) for i in range(36, 76)}  # (v45 .. v47, v63: FS_FAST_LOOP_FD16P's checkpoint and deferred verdicts)
FIRST = re.compile(r"^v_pk_fma_f32 v\[56:57\], v\[48:49\], v\[\d+:\d+\], s\[6(4:65|6:67)\]")


def successors(instrs):
    index = {ins["addr"]: i for i, ins in enumerate(instrs)}
    succ = []
    for i, ins in enumerate(instrs):
        nxt = [i + 1] if i + 1 < len(instrs) else []
        if ins["op"].startswith(chk.BRANCHES):
            imm = int(ins["ops"].split()[0])
            imm = imm - 65536 if imm >= 32768 else imm
            t = index.get(ins["addr"] + 4 + imm * 4)
            tgt = [t] if t is not None else []
            succ.append(tgt if ins["op"] == "s_branch" else nxt + tgt)
        elif ins["op"] in ("s_endpgm", "s_setpc_b64"):
            succ.append([])
        else:
            succ.append(nxt)
    return succ


def defs_uses(ins):
    """(registers written, registers read) of one instruction, conservatively: the first operand is the destination of every
    instruction that has one; stores, compares into vcc, branches, waits and s_cmp have none."""
    op, ops = ins["op"], ins["ops"]
    no_dst = op.startswith(("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "s_cmp", "s_cbranch",
                            "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_setreg", "s_barrier", "s_bitcmp", "v_cmpx",
                            "global_atomic", "ds_add", "s_sleep", "s_setprio", "s_sendmsg", "buffer_wbl2", "buffer_inv"))
    if no_dst:
        return set(), chk.regs_of(ops)
    dst, _, src = ops.partition(",")
    d = chk.regs_of(dst)
    u = chk.regs_of(src)
    if op.startswith("v_pk_") and "op_sel" in src:
        # a packed operation reads, of each 64-bit source, only the halves op_sel (for the result's low half: 0 = low register)
        # and op_sel_hi (for its high half: 1 = high register, the default) name -- `s[36:37] op_sel_hi:[1,0]` reads s36 twice
        # and s37 not at all
        m_lo = re.search(r"op_sel:\[([01,]+)\]", src)
        m_hi = re.search(r"op_sel_hi:\[([01,]+)\]", src)
        srcs = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", src.split(" op_sel")[0].split(" neg_")[0]) if x.strip()]
        lo = [int(x) for x in m_lo.group(1).split(",")] if m_lo else [0] * len(srcs)
        hi = [int(x) for x in m_hi.group(1).split(",")] if m_hi else [1] * len(srcs)
        u = set()
        for k, operand in enumerate(srcs):
            regs = sorted(chk.regs_of(operand))
            if len(regs) == 2 and k < len(lo) and k < len(hi):
                u |= {regs[h] for h in {lo[k], hi[k]}}
            else:
                u |= set(regs)
    if op.startswith(("v_readlane", "v_writelane", "v_mac", "v_fmac", "v_pk_fmac", "v_dot", "v_mfma", "v_cndmask")) or "_mov_rel" in op:
        u |= d if op.startswith(("v_writelane", "v_mac", "v_fmac", "v_pk_fmac")) else set()
    if op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64_u32", "v_mad_i64_i32")):
        # two destinations: vdst, sdst
        second, _, rest = src.partition(",")
        d |= chk.regs_of(second)
        u = chk.regs_of(rest)
    return d, u


def regions(instrs):
    """[(first index, last index, scratch set)] of the hand-scheduled loops of a function."""
    out = []
    i = 0
    while i < len(instrs):
        if FIRST.match(instrs[i]["text"]):
            # back to the loop's top: the nearest preceding write of v62 from v60 (block test) or of v61 (deferred form)
            # (the pipelined 16-step body, FS_FAST_LOOP_FD16P, resets v61 a second time at its per-body checkpoint, right in front
            # of the first packed instruction: its top is the EARLIER of the two, in front of the statement's first loads)
            a = i
            while a > 0 and i - a < 12 and not (instrs[a]["text"].startswith(("v_max_i32_e32 v62, v60", "v_mov_b32_e32 v61, 0x7f800000"))):
                a -= 1
            if instrs[a]["text"].startswith("v_mov_b32_e32 v61, 0x7f800000") and "s[66:67]" in instrs[i]["text"]:
                k = a - 1
                while k > 0 and a - k < 28:
                    if instrs[k]["text"].startswith("v_mov_b32_e32 v61, 0x7f800000"):
                        a = k
                        break
                    k -= 1
            # forward to the common end: the first s_waitcnt lgkmcnt(0) that no branch of the region jumps over
            b = i
            last_target = i
            index = {ins["addr"]: k for k, ins in enumerate(instrs)}
            for k in range(a, i):  # (the branches of the loop's top -- the verdict, the first block's test -- leave forwards too)
                ins = instrs[k]
                if ins["op"].startswith(chk.BRANCHES):
                    imm = int(ins["ops"].split()[0])
                    imm = imm - 65536 if imm >= 32768 else imm
                    t = index.get(ins["addr"] + 4 + imm * 4)
                    if t is not None and t > i:
                        last_target = max(last_target, t)
            while b + 1 < len(instrs):
                b += 1
                ins = instrs[b]
                if ins["op"].startswith(chk.BRANCHES):
                    imm = int(ins["ops"].split()[0])
                    imm = imm - 65536 if imm >= 32768 else imm
                    t = index.get(ins["addr"] + 4 + imm * 4)
                    if t is not None and t > b:
                        last_target = max(last_target, t)
                if ins["op"] == "s_waitcnt" and "lgkmcnt(0)" in ins["ops"] and b >= last_target and b - i > 60:
                    break
            out.append((a, b, SCRATCH16 if "s[66:67]" in instrs[i]["text"] else SCRATCH))
            i = b + 1
        else:
            i += 1
    return out


def check_function(name, instrs):
    findings = []
    regs = regions(instrs)
    if not regs:
        return findings, 0
    succ = successors(instrs)
    du = [defs_uses(ins) for ins in instrs]
    # the loop's own reads of its scratch registers are not uses the surrounding code is responsible for (e.g. `s_mov eb, s62`
    # at the loop's head reads what the previous body loaded -- or, on entry, a value nothing depends on)
    inside = {}
    for a, b, scr in regs:
        for i in range(a, b + 1):
            inside[i] = scr
    du = [(d, (u - inside[i]) if i in inside else u) for i, (d, u) in enumerate(du)]
    live_in = [set() for _ in instrs]
    changed = True
    while changed:  # backward liveness, to a fixed point
        changed = False
        for i in range(len(instrs) - 1, -1, -1):
            out = set()
            for j in succ[i]:
                out |= live_in[j]
            d, u = du[i]
            new = u | (out - d)
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    for a, b, scr in regs:
        # exits of the region: successors outside [a, b]
        live_out = set()
        for i in range(a, b + 1):
            for j in succ[i]:
                if j < a or j > b:
                    live_out |= live_in[j]
        bad = sorted(live_out & scr)
        if bad:
            findings.append((name, "scratch registers of the hand-scheduled loop live on its exit", bad))
        written = set()
        for i in range(a, b + 1):
            written |= du[i][0]
        foreign_v = sorted(r for r in written if r[0] == "v" and r not in NAMED)
        foreign_s = sorted(r for r in written if r[0] == "s" and r not in NAMED)
        if foreign_v:
            findings.append((name, "the loop writes vector registers it does not name", foreign_v))
        if len(foreign_s) > 12:
            findings.append((name, "the loop writes more scalar registers than it has operands", foreign_s))
    return findings, len(regs)


def run(so_path):
    funcs = chk.disassemble(so_path)
    findings, n = [], 0
    for name, lines in funcs.items():
        if "k_lav2_hdr32_fast" not in name and "k_perturb_scalar" not in name:
            continue
        f, k = check_function(name, lines)
        findings += f
        n += k
    return findings, n


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(chk.ROOT, "fractalshark_amd", "csrc", "libfsmi355.so")
    findings, n = run(so)
    print("%d hand-scheduled loops checked" % n)
    for name, what, regs in findings:
        print("FINDING %s\n    %s: %s" % (name, what, regs))
    sys.exit(1 if findings else 0)
