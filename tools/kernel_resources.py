#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS / occupancy of every kernel in libfsmi355.so (read from the gfx950 code objects'
metadata notes; CPU-only).  Usage: python tools/kernel_resources.py [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.join(ROOT, "fractalshark_amd", "csrc", "libfsmi355.so")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    import shutil
    shutil.copy(lib, os.path.join(d, "lib.so"))  # llvm-objdump --offloading extracts NEXT TO its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.join(d, "lib.so")], cwd=d, check=True,
                   stdout=subprocess.DEVNULL)
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readobj"), "--notes", os.path.join(d, f)], stdout=subprocess.PIPE,
                               text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
            name = g("name")
            try:
                name = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], stdout=subprocess.PIPE, text=True).stdout.strip()
            except OSError:
                pass
            if flt and flt not in name:
                continue
            v = int(g("vgpr_count"))
            occ = min(8, 512 // max(1, (v + 7) // 8 * 8))
            print("%-90s vgpr %3d sgpr %3s scratch %4s lds %5s waves/SIMD %d" % (name[:90], v, g("sgpr_count"),
                                                                                g("private_segment_fixed_size"),
                                                                                g("group_segment_fixed_size"), occ))
