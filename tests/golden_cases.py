"""The reference's twelve golden render cases (FractalSharkTest/TestRenderGoldens.cpp:84-97) as data: name, built-in
view, CPU RenderAlgorithm, antialiasing and the CRC-64 literal of the 256x256 PNG, plus how each case is fed to the
oracle (CPU) and to the HIP path (GPU).  Shared by tests/test_gpu_goldens.py and tests/golden/make_golden_crc.py."""
import zlib

import numpy as np

W = H = 256  # kGoldenWidth / kGoldenHeight

# (case name, view, CPU RenderAlgorithm, AA, CRC-64 of the PNG)
CASES = [
    ("view0-cpu64", 0, "Cpu64", 1, "1275500d639ad02e"),
    ("view0-cpu64-aa4", 0, "Cpu64", 4, "39671027bacf2567"),
    ("view1-cpu-bla", 1, "Cpu64PerturbedBLAHDR", 1, "d0c8921c878f6dc3"),
    ("view0-cpuhdr", 0, "CpuHDR32", 1, "66ba2caaaa7f8013"),
    ("view5-cpu-bla-v2", 5, "Cpu32PerturbedBLAV2HDR", 1, "1233a56b293e7b08"),
    ("view0-cpuhdr64", 0, "CpuHDR64", 1, "1275500d639ad02e"),
    ("view5-cpu-perturbed-bla", 5, "Cpu64PerturbedBLA", 1, "f201db00ade569fc"),
    ("view5-cpu32-bla-hdr", 5, "Cpu32PerturbedBLAHDR", 1, "634d826801d54979"),
    ("view5-cpu64-bla-hdr", 5, "Cpu64PerturbedBLAHDR", 1, "c91e33c3eb85b33d"),
    ("view5-cpu64-bla-v2", 5, "Cpu64PerturbedBLAV2HDR", 1, "ca7ad7c5f9cf750e"),
    ("view5-cpu32-rc-bla-v2", 5, "Cpu32PerturbedRCBLAV2HDR", 1, "b956600cfdfe431a"),
    ("view5-cpu64-rc-bla-v2", 5, "Cpu64PerturbedRCBLAV2HDR", 1, "68df9ceecaf1a667"),
]


def buffer_crc32(iters):
    """CRC-32 of the padded uint32 iteration buffer (committed in tests/golden/golden_crc.json for every case)."""
    return "%08x" % (zlib.crc32(np.ascontiguousarray(iters, np.uint32).tobytes()) & 0xFFFFFFFF)


def build_inputs(inputs, view_n, alg, aa):
    """Host inputs of one case: (view, orbit-or-None, table-or-None)."""
    v = inputs.View.builtin(view_n, W, H, antialiasing=aa)
    if alg in ("Cpu64", "CpuHDR32", "CpuHDR64"):
        return v, None, None
    if alg == "Cpu64PerturbedBLA":
        return v, inputs.OrbitF64(v), None
    is64 = alg.startswith("Cpu64")
    rc = "RC" in alg
    ob = inputs.Orbit(v, is64=is64, compression_exp=20 if rc else None)
    table = inputs.LATable(ob) if "V2" in alg else inputs.BLATable(ob)
    return v, ob, table


def oracle_render(_oracle, alg, v, ob, table, aa):
    if alg == "Cpu64":
        return _oracle.direct_f64(v, aa=aa)
    if alg in ("CpuHDR32", "CpuHDR64"):
        return _oracle.direct_hdr(v, alg == "CpuHDR64")
    if alg == "Cpu64PerturbedBLA":
        return _oracle.bla_f64(v, ob)
    if "V2" in alg:
        return _oracle.lav2_hdr32(v, ob, table, stage_test=0)
    return _oracle.bla_hdr32(v, ob, table)
