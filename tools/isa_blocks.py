#!/usr/bin/env python3
"""Basic blocks of one kernel of libfsmi355.so with their instruction mix (CPU-only; reads the gfx950 code objects).

For every basic block: start address, instructions, vector ALU, scalar ALU, memory instructions, terminator and branch
target -- the static side of the per-phase cycle probes (tools/c5_phase_probe.py, tools/cycle_probe.py): what one pass of
a wave through a phase issues.  Usage: python tools/isa_blocks.py '<substring of the mangled or demangled kernel name>'
       e.g. 'k_perturb_scalar<float, true, false, false>'"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels():
    """-> {demangled name: [(address, opcode, operands), ...]} for every function of every gfx950 code object."""
    lib = os.path.join(ROOT, "fractalshark_amd", "csrc", "libfsmi355.so")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True,
                       stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", f], cwd=d,
                                  stdout=subprocess.PIPE, text=True, check=True).stdout
            cur = None
            for ln in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
                if m:
                    cur = out.setdefault(m.group(1), [])
                    continue
                m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
                if m and cur is not None:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def blocks(ins):
    is_br = lambda op: op.startswith("s_cbranch") or op == "s_branch"  # noqa: E731

    def target(a, ops):
        off = int(ops.split()[-1])
        return a + 4 + (off - 65536 if off >= 32768 else off) * 4

    index = {a: i for i, (a, _, _) in enumerate(ins)}
    starts = {ins[0][0]}
    for i, (a, op, ops) in enumerate(ins):
        if is_br(op):
            starts.add(target(a, ops))
        if (is_br(op) or op == "s_endpgm") and i + 1 < len(ins):
            starts.add(ins[i + 1][0])
    starts = sorted(x for x in starts if x in index)
    for k, st in enumerate(starts):
        i, e = index[st], (index[starts[k + 1]] if k + 1 < len(starts) else len(ins))
        ops = [ins[j][1] for j in range(i, e)]
        last = ins[e - 1]
        yield {"start": st, "n": e - i, "valu": sum(o.startswith("v_") for o in ops),
               "salu": sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer")) for o in ops),
               "mem": sum(o.startswith(("global_", "flat_", "buffer_", "ds_", "s_load", "s_buffer", "scratch_")) for o in ops),
               "end": last[1], "target": target(last[0], last[2]) if is_br(last[1]) else None}


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_perturb_scalar<float, true, false, false>"
    ks = kernels()
    hits = [n for n in ks if want in n]
    if not hits:
        sys.exit("no kernel matches %r; have e.g. %s" % (want, sorted(ks)[:5]))
    for name in hits:
        ins = ks[name]
        bl = list(blocks(ins))
        print("%s: %d instructions, %d vector, %d blocks" % (name, len(ins), sum(b["valu"] for b in bl), len(bl)))
        for b in bl:
            print("  %#8x  n %4d  valu %4d  salu %3d  mem %2d  %-18s %s" % (b["start"], b["n"], b["valu"], b["salu"], b["mem"],
                                                                            b["end"], "-> %#x" % b["target"] if b["target"] else ""))


if __name__ == "__main__":
    main()
