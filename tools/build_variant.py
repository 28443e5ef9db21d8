#!/usr/bin/env python3
"""Build an A/B variant of libfsmi355.so: the named translation units recompiled with extra -D flags, linked with the product's
other objects, into build/ab/libfsmi355_<name>.so (git-ignored, but it travels to the GPU box with the snapshot).  Select it at
run time with FSMI355_LIB=build/ab/libfsmi355_<name>.so (fractalshark_amd/_capi.py prints the path it loads).

  python tools/build_variant.py <name> <unit.hip>[,<unit2.hip>] -DFOO=1 [-DBAR ...]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fractalshark_amd import _build  # noqa: E402


def main():
    name, units, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    # the product objects the variant links with: built by the product's own recipe (never under the variant's environment --
    # FS_PEEPHOLE=0 for an A/B of the peephole would otherwise rebuild the PRODUCT without it)
    env = {k: v for k, v in os.environ.items() if not k.startswith("FS_")}  # (FS_PEEPHOLE_UNITS, FS_VERIFY_* ... belong to the variant)
    subprocess.run([sys.executable, "-c", "from fractalshark_amd import _build; _build.build_render()"], check=True, env=env, cwd=ROOT)
    out_dir = os.path.join(ROOT, "build", "ab", name)
    os.makedirs(out_dir, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    for src in _build._render_units():
        base = os.path.basename(src)
        if base in units:
            obj = os.path.join(out_dir, base + ".o")
            _build.compile_one(hipcc, src, obj, [*_build._render_flags(), *_build._UNIT_FLAGS.get(base, []), *defs],
                               _build._peephole_on(src))
        else:
            obj = os.path.join(_build.OBJ, base + ".o")
        objs.append(obj)
    lib = os.path.join(ROOT, "build", "ab", "libfsmi355_%s.so" % name)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl", "-lpthread"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
