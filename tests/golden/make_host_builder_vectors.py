#!/usr/bin/env python3
"""Values the reference's own unit tests hold for two host-side inputs of the hot path -> tests/golden/host_builder_vectors.json.

Source (read where it lies, /root/reference; nothing of it is copied):
  FractalSharkTest/TestLAParameters.cpp      LAParam_DefaultConstruction / LAParam_DefaultExponents: the defaults every LAv2 table
                                             is built with (detection method and the six threshold exponents)
  FractalSharkTest/TestPrecisionCalculator.cpp   the bounding-box and converter cases: the MPIR precision a view is given
                                             (larger binary exponent of its width / height + 120 bits), which decides every bit of the
                                             reference orbit
  FractalSharkTest/TestPointZoomBBConverter.cpp  SquareAspectRatio_AlreadySquare / _Wide / _Tall: how a view's box is widened to the
                                             screen's aspect ratio before anything is rendered (every pixel's delta-c follows from it)
  FractalSharkTest/TestCudaDblflt.cpp        the double -> (head, tail) split of MattDblflt / CudaDblflt (construction from a double,
                                             copy, assignment): the conversion every HDRFloat<CudaDblflt> and CudaDblflt input goes through
What is committed is DATA: getter names with expected integers, boxes (as decimal / power-of-two text) with the expected
precision, and doubles with the tolerance their head + tail must reproduce them to.  Usage: python tests/golden/make_host_builder_vectors.py"""
import json
import os
import re

REF = "/root/reference/FractalSharkTest"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_builder_vectors.json")


def body_of(text, name):
    m = re.search(r"TEST\(%s\)\s*\{(.*?)\n\}" % name, text, re.S)
    return m.group(1), text[:m.start()].count("\n") + 1


def main():
    la = open(os.path.join(REF, "TestLAParameters.cpp")).read()
    params = {}
    src = []
    for name in ("LAParam_DefaultConstruction", "LAParam_DefaultExponents"):
        body, line = body_of(la, name)
        src.append("TestLAParameters.cpp:%d" % line)
        for getter, value in re.findall(r"ASSERT_EQ\(p\.(Get\w+)\(\),\s*(-?\d+)\)", body):
            params[getter] = int(value)
    pc = open(os.path.join(REF, "TestPrecisionCalculator.cpp")).read()
    extra = int(re.search(r"MinExtra\s*=\s*(\d+)", pc).group(1))
    boxes = []
    for name in ("PC_FromBoundingBox", "PC_FromBoundingBox_AsymmetricDeltas"):
        body, line = body_of(pc, name)
        nums = re.findall(r"HighPrecision\{(-?[\d.]+)\}", body)
        want = int(re.search(r"ASSERT_EQ\(prec,\s*(\d+)\s*\+\s*MinExtra\)", body).group(1)) + extra
        boxes.append({"name": name, "source": "TestPrecisionCalculator.cpp:%d" % line, "box": nums[:4], "precision_bits": want})
    # the converter cases: centre (0, 0) and a zoom z span the box [-2/z, 2/z]^2 (PointZoomBBConverter, Factor 2)
    body, line = body_of(pc, "PC_FromConverter_DefaultZoom")
    want = int(re.search(r"ASSERT_EQ\(prec,\s*(\d+)\s*\+\s*MinExtra\)", body).group(1)) + extra
    boxes.append({"name": "PC_FromConverter_DefaultZoom", "source": "TestPrecisionCalculator.cpp:%d" % line,
                  "box": ["-2", "-2", "2", "2"], "precision_bits": want})
    body, line = body_of(pc, "PC_FromConverter_DeepZoom")
    doublings = int(re.search(r"for \(int i = 0; i < (\d+);", body).group(1))
    want = int(re.search(r"ASSERT_EQ\(prec,\s*(\d+)\s*\+\s*MinExtra\)", body).group(1)) + extra
    boxes.append({"name": "PC_FromConverter_DeepZoom", "source": "TestPrecisionCalculator.cpp:%d" % line,
                  "box_pow2": {"half_width_exp": 1 - doublings}, "precision_bits": want})
    # double -> double-float: every case that builds a MattDblflt / CudaDblflt from a double literal and checks what comes back
    df = open(os.path.join(REF, "TestCudaDblflt.cpp")).read()
    splits = []
    for m in re.finditer(r"TEST\((Dblflt_\w+)\)\s*\{(.*?)\n\}", df, re.S):
        name, body = m.group(1), m.group(2)
        line = df[:m.start()].count("\n") + 1
        c = (re.search(r"(?:MattDblflt|CDf) \w+\((-?\d+\.\d+)\);", body) or re.search(r"double \w+ = (-?\d+\.\d+);", body) or
             re.search(r"\bcd = (-?\d+\.\d+);", body))
        if not c or "HighPrecision" in body or "ToString" in body:
            continue
        value = c.group(1)
        tol = None
        for got, want, t in re.findall(r"ASSERT_NEAR\(([^,]+),\s*([^,]+),\s*([-\d.e]+)f?\)", body):
            if "static_cast<double>" in got or got.strip() in ("reconstructed", "back") or want.strip() in (value, "original"):
                if "head" in got or "tail" in got:
                    continue
                tol = float(t) if tol is None else min(tol, float(t))
        entry = {"name": name, "source": "TestCudaDblflt.cpp:%d" % line, "value": value}
        if tol is not None:
            entry["sum_tolerance"] = tol
        if name.endswith("_Zero"):
            entry["both_words_zero"] = True
        if name.endswith("PrecisionGain"):
            entry["closer_than_the_head_alone"] = True
        if len(entry) > 3:
            splits.append(entry)
    pz = open(os.path.join(REF, "TestPointZoomBBConverter.cpp")).read()
    aspect = []
    for name in ("SquareAspectRatio_AlreadySquare", "SquareAspectRatio_Wide", "SquareAspectRatio_Tall"):
        body, line = body_of(pz, name)
        box = re.findall(r"HighPrecision\{(-?\d+)\}", body)[:4]
        w, h = re.search(r"SquareAspectRatio\((\d+),\s*(\d+)\)", body).groups()
        want = {}
        for got, val, tol in re.findall(r"ASSERT_NEAR\((\w+),\s*([\w.]+),\s*([-\de.]+)\)", body):
            key = "width" if "idth" in got else "height"
            if re.fullmatch(r"[\d.]+", val):
                want[key] = (float(val), float(tol))
            else:  # compared with the box's own extent before the call
                want[key] = (float(box[2 if key == "width" else 3]) - float(box[0 if key == "width" else 1]), float(tol))
        aspect.append({"name": name, "source": "TestPointZoomBBConverter.cpp:%d" % line, "box": box, "screen": [int(w), int(h)],
                       "width": want["width"][0], "height": want["height"][0], "tolerance": max(want["width"][1], want["height"][1])})
    out = {"_comment": "generated by tests/golden/make_host_builder_vectors.py from the reference's unit tests "
                       "(TestLAParameters.cpp, TestPrecisionCalculator.cpp, TestPointZoomBBConverter.cpp, TestCudaDblflt.cpp): names, "
                       "operands, expected values and tolerances only",
           "la_parameters": {"sources": src, "defaults": params}, "precision": boxes, "square_aspect_ratio": aspect,
           "double_float_split": splits}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
