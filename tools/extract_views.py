#!/usr/bin/env python3
"""Extract the built-in view presets (coordinate strings, iteration count, antialiasing) that the
BASELINE configs use from the reference's FractalSharkLib/FractalViewPresets.cpp into
fractalshark_amd/data/views.json.

The presets are *input data* of the hot path (decimal coordinate strings); nothing else is taken from
the file.  Run in the build container only (/root/reference is not present on the GPU box).
"""
import json
import re
import sys

SRC = "/root/reference/FractalSharkLib/FractalViewPresets.cpp"
WANT = [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 14, 15, 17, 19, 22]


def main():
    text = open(SRC, encoding="utf-8-sig").read()
    out = {
        # case 0 (FractalViewPresets.cpp:2036-2052): centre (0,0), zoom 1 -> [-2,2]^2, default iterations
        "0": {"minX": "-2", "minY": "-2", "maxX": "2", "maxY": "2", "numIterations": 8192, "gpuAntialiasing": 1,
              "from_point_zoom": True},
    }
    for v in WANT:
        m = re.search(r"case %d:\s*\{?(.*?)break;" % v, text, re.S)
        body = m.group(1)
        entry = {"gpuAntialiasing": 1}
        for key in ("minX", "minY", "maxX", "maxY"):
            mm = re.search(r"result\.%s\s*=\s*HighPrecision\{(.*?)\};" % key, body, re.S)
            lits = re.findall(r'"([^"]*)"', mm.group(1))
            entry[key] = "".join(lits)
        mm = re.search(r"result\.numIterations\s*=\s*([0-9']+);", body)
        entry["numIterations"] = int(mm.group(1).replace("'", ""))
        mm = re.search(r"result\.gpuAntialiasing\s*=\s*([0-9]+);", body)
        if mm:
            entry["gpuAntialiasing"] = int(mm.group(1))
        out[str(v)] = entry
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else
                        "/root/repo/fractalshark_amd/data/views.json", "w"), indent=1)
    for k, e in out.items():
        print(k, len(e["minX"]), e["numIterations"], e["gpuAntialiasing"])


if __name__ == "__main__":
    main()
