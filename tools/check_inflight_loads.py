"""Static check of the built gfx950 code objects: no instruction touches the destination registers of a load that may
still be in flight.

The tuned loops issue their orbit loads by hand (`asm volatile("s_load_dwordx4 ...")`, `global_load_dwordx3`) and wait
for them with a separate `s_waitcnt` statement tied to the destination operands.  The compiler knows nothing about the
hardware's counters for those loads: if its register allocator or scheduler placed a copy of (or a write to) such a
register between the load and the wait, the kernel would read stale data or have its registers overwritten late --
silently.  This script disassembles libfsmi355.so and walks every function linearly:

  * scalar loads (s_load_* / s_buffer_load_*) return out of order: their destinations are in flight until
    `s_waitcnt lgkmcnt(0)`;
  * vector memory loads return in order: `s_waitcnt vmcnt(N)` retires all but the youngest N (stores count too);
  * any other instruction that names an in-flight register, as source or destination, is reported.

It is a forward may-analysis over each function's control-flow graph (branch targets decoded from the simm16 offsets),
the same conservative model the compiler's own wait-count insertion uses -- so compiler-issued loads pass by
construction and a finding means a hand-issued load was mishandled.  Exit status 1 on a finding.
Usage: python tools/check_inflight_loads.py [path/to/libfsmi355.so]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([sv])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            for i in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((kind, i))
    return out


def waitcnt(ops):
    """-> (vmcnt or None, lgkmcnt or None) of an s_waitcnt operand string."""
    vm = re.search(r"vmcnt\((\d+)\)", ops)
    lg = re.search(r"lgkmcnt\((\d+)\)", ops)
    return (int(vm.group(1)) if vm else None, int(lg.group(1)) if lg else None)


BRANCHES = ("s_branch", "s_cbranch_")
VM_LOAD = ("global_load_", "buffer_load_", "flat_load_", "scratch_load_")
VM_OTHER = ("global_store_", "buffer_store_", "flat_store_", "scratch_store_", "global_atomic_", "buffer_atomic_",
            "flat_atomic_")


def step(state, ins, report=None):
    """Transfer function of one instruction; state = (frozenset of (register, issuing address) pairs of the scalar loads
    in flight, tuple of such frozensets: the vmcnt-counted operations still outstanding, oldest first)."""
    scalar, vector = state
    op, ops = ins["op"], ins["ops"]
    if op == "s_waitcnt":
        vm, lg = waitcnt(ops)
        if lg == 0:
            scalar = frozenset()
        if vm is not None:
            vector = vector[max(0, len(vector) - vm):]
        return scalar, vector
    if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
        return frozenset(), ()
    is_load, is_vm_other = op.startswith(VM_LOAD), op.startswith(VM_OTHER)
    # LDS-DMA (global_load_lds_*, buffer_load_* ... lds): counted by vmcnt like a load, but its destination is LDS -- every
    # register operand is an address that is read at issue, and no register is written when it lands
    is_lds_dma = op.startswith("global_load_lds_") or (op.startswith("buffer_load_") and ops.rstrip().endswith(" lds"))
    if is_lds_dma:
        is_load, is_vm_other = False, True
    dst_text, _, src_text = ops.partition(",")
    # vector memory operations complete in order: a load may write a register an older load also writes; only its
    # address operands must have arrived
    used = regs_of(src_text) if is_load else regs_of(ops)
    busy = {}
    for r, a in scalar:
        busy[r] = a
    for dst in vector:
        for r, a in dst:
            busy[r] = a
    if report is not None:
        hit = sorted(r for r in used if r in busy)
        if hit:
            report.append((ins["text"], [(r, hex(busy[r])) for r in hit]))
    here = ins["addr"]
    if op.startswith(("s_load_", "s_buffer_load_")):
        scalar = scalar | frozenset((r, here) for r in regs_of(dst_text))
    elif is_load:
        vector = vector + (frozenset((r, here) for r in regs_of(dst_text)),)
    elif is_vm_other:
        returns = not is_lds_dma and "_atomic_" in op and (" glc" in ops or " sc0" in ops)
        vector = vector + (frozenset((r, here) for r in regs_of(dst_text)) if returns else frozenset(),)
    return scalar, vector


def merge(a, b):
    """Join of two states: a register is in flight if it is on either path; vmcnt(N) retires all but the youngest N, so
    the FIFOs are aligned at their young ends."""
    if a is None:
        return b
    sa, va = a
    sb, vb = b
    n = max(len(va), len(vb))
    va = (frozenset(),) * (n - len(va)) + va
    vb = (frozenset(),) * (n - len(vb)) + vb
    return sa | sb, tuple(x | y for x, y in zip(va, vb))


def check_function(name, instrs):
    """instrs: list of dicts {addr, size, op, ops, text}.  Forward may-analysis over the control-flow graph."""
    if not instrs:
        return []
    index = {ins["addr"]: i for i, ins in enumerate(instrs)}
    succ = []
    for i, ins in enumerate(instrs):
        nxt = [i + 1] if i + 1 < len(instrs) else []
        if ins["op"].startswith(BRANCHES):
            imm = int(ins["ops"].split()[0])
            imm = imm - 65536 if imm >= 32768 else imm
            t = index.get(ins["addr"] + 4 + imm * 4)
            tgt = [t] if t is not None else []
            succ.append(tgt if ins["op"] == "s_branch" else nxt + tgt)
        elif ins["op"] in ("s_endpgm", "s_setpc_b64"):
            succ.append([])
        else:
            succ.append(nxt)
    state_in = [None] * len(instrs)
    state_in[0] = (frozenset(), ())
    work = [0]
    while work:
        i = work.pop()
        out = step(state_in[i], instrs[i])
        out = (out[0], out[1][-64:])
        for j in succ[i]:
            m = merge(state_in[j], out)
            if m != state_in[j]:
                state_in[j] = m
                work.append(j)
    findings = []
    for i, ins in enumerate(instrs):
        if state_in[i] is None:
            continue
        rep = []
        step(state_in[i], ins, rep)
        findings += [(name, t, r) for t, r in rep]
    return findings


def disassemble(so_path):
    tmp = tempfile.mkdtemp(prefix="fs_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = f.split(".hipv4")[0] + ":" + m.group(1)
                    funcs[cur] = []
                    continue
                if cur is None or not line.startswith("\t") or "//" not in line:
                    continue
                body, comment = line.split("//", 1)
                body = body.strip()
                cm = re.match(r"\s*([0-9A-Fa-f]+):((?:\s+[0-9A-Fa-f]{8})+)", comment)
                if not body or not cm:
                    continue
                parts = body.split(None, 1)
                funcs[cur].append({"addr": int(cm.group(1), 16), "size": 4 * len(cm.group(2).split()), "op": parts[0],
                                   "ops": parts[1] if len(parts) > 1 else "", "text": body})
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run(so_path):
    funcs = disassemble(so_path)
    findings = []
    n_scalar = n_vector = 0
    for name, lines in funcs.items():
        n_scalar += sum(1 for l in lines if l["op"].startswith("s_load_"))
        n_vector += sum(1 for l in lines if l["op"].startswith("global_load_"))
        findings += check_function(name, lines)
    return funcs, findings, n_scalar, n_vector


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fractalshark_amd", "csrc", "libfsmi355.so")
    funcs, findings, n_scalar, n_vector = run(so)
    print(f"{len(funcs)} functions, {n_scalar} scalar loads, {n_vector} global loads checked")
    for name, ln, regs in findings:
        print(f"IN-FLIGHT REGISTER USED  {name}\n    {ln}\n    registers: {regs}")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
