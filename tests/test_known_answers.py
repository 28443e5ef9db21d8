"""The reference's known-answer unit vectors for its numeric types, run against this repository's arithmetic.

tests/golden/known_answer_vectors.json holds, as data, the straight-line programs and expected values of the reference's
FractalSharkTest/Test{HDRFloat,HDRFloatComplex,ATInfo,BLA}.cpp (written by tests/golden/make_known_answer_vectors.py, which
reads the reference where it lies).  tests/kat/kat_vm.hpp maps every operation name onto csrc/hdr_math.hpp, bla_math.hpp and
at_math.hpp and evaluates the assertions with the reference's own tolerances (ASSERT_NEAR) -- once in a g++ build of the
headers (CPU test) and once inside one gfx950 kernel (`-m gpu`), where PerformAT runs the kernels' tuned at_perform."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = os.path.join(HERE, "kat")
VECTORS = os.path.join(HERE, "golden", "known_answer_vectors.json")
CSRC = os.path.join(os.path.dirname(HERE), "fractalshark_amd", "csrc")
INSTR = np.dtype([("op", "<i4"), ("a", "<i4"), ("n", "<i4"), ("pad", "<i4"), ("d", "<f8")])
CHECK = np.dtype([("ok", "<i4"), ("kind", "<i4"), ("got", "<f8"), ("want", "<f8"), ("tol", "<f8")])
OPS = {"push_int": 0, "push_dbl": 1, "push_flt": 2, "load": 3, "store": 4, "call": 5, "assert_near": 6, "assert_eq": 7,
       "assert_true": 8, "assert_false": 9}


def _func_ids():
    src = open(os.path.join(KAT, "kat_vm.hpp")).read()
    block = src[src.index("#define KAT_FUNCS(X)"):src.index("enum Func {")]
    return {name: i for i, name in enumerate(re.findall(r"X\((\w+)\)", block))}


def _stale(target, sources):
    return not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in sources)


def _sources(main):
    return [os.path.join(KAT, main), os.path.join(KAT, "kat_vm.hpp")] + \
           [os.path.join(CSRC, h) for h in ("hdr_math.hpp", "bla_math.hpp", "at_math.hpp")]


def _build_host():
    lib = os.path.join(KAT, "libkat_host.so")
    if _stale(lib, _sources("kat_host.cpp")):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", lib,
                        os.path.join(KAT, "kat_host.cpp")], check=True)
    return C.CDLL(lib)


def _build_device():
    lib = os.path.join(KAT, "libkat_device.so")
    if _stale(lib, _sources("kat_device.hip")):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-shared",
                        "-fPIC", "-o", lib, os.path.join(KAT, "kat_device.hip")], check=True)
    return C.CDLL(lib)


def _encode(case, ids):
    prog = np.zeros(len(case["ops"]), INSTR)
    for k, op in enumerate(case["ops"]):
        prog[k]["op"] = OPS[op[0]]
        if op[0] in ("push_int", "push_dbl", "push_flt"):
            prog[k]["d"] = float(op[1])
        elif op[0] in ("load", "store"):
            prog[k]["a"] = op[1]
        elif op[0] == "call":
            prog[k]["a"] = ids[op[1]]
            prog[k]["n"] = op[2]
    return prog


def _run_all(fn):
    data = json.load(open(VECTORS))
    ids = _func_ids()
    used = {op[1] for c in data["cases"] for op in c["ops"] if op[0] == "call"}
    assert used <= set(ids), "operations without an implementation in kat_vm.hpp: %s" % sorted(used - set(ids))
    assert len(data["cases"]) >= 80
    failures, total = [], 0
    for case in data["cases"]:
        prog = _encode(case, ids)
        checks = np.zeros(case["asserts"], CHECK)
        n = fn(prog.ctypes.data, len(prog), checks.ctypes.data, len(checks))
        assert n == case["asserts"], "%s: %d assertions evaluated, %d expected" % (case["name"], n, case["asserts"])
        total += n
        for k, c in enumerate(checks):
            if not c["ok"]:
                failures.append("%s (%s) assertion %d: got %r, want %r, tol %r" %
                                (case["name"], case["source"], k, float(c["got"]), float(c["want"]), float(c["tol"])))
    assert not failures, "\n".join(failures)
    return len(data["cases"]), total


def test_vectors_are_data_and_cover_the_reference_files():
    data = json.load(open(VECTORS))
    src = {c["source"].split(":")[0] for c in data["cases"]}
    assert src == {"TestHDRFloat.cpp", "TestHDRFloatComplex.cpp", "TestATInfo.cpp", "TestBLA.cpp", "TestFloatComplex.cpp"}
    # operands and operation names only: no statement of the reference's source text
    blob = open(VECTORS).read()
    assert "ASSERT_" not in blob and "HDRd " not in blob and ";" not in blob
    for s in data["skipped"]:
        assert s["why"]


def test_vectors_regenerate_identically_where_the_reference_is_present(tmp_path):
    if not os.path.isdir("/root/reference/FractalSharkTest"):
        pytest.skip("the reference tree is not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mkv", os.path.join(HERE, "golden", "make_known_answer_vectors.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.OUT = str(tmp_path / "v.json")
    m.main()
    assert json.load(open(m.OUT)) == json.load(open(VECTORS))


def test_known_answers_host_build():
    lib = _build_host()
    lib.kat_run_host.restype = C.c_int
    lib.kat_run_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    cases, checks = _run_all(lib.kat_run_host)
    assert cases >= 80 and checks >= 160


@pytest.mark.gpu
def test_known_answers_on_the_device():
    lib = _build_device()
    lib.kat_run_device.restype = C.c_int
    lib.kat_run_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    cases, checks = _run_all(lib.kat_run_device)
    assert cases >= 80 and checks >= 160
