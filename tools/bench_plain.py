"""Times the non-HDR LAv2 kernels (Gpu1x32 / Gpu1x64 / Gpu2x32 PerturbedLAv2*) on shallow views at 3840x2160.
Usage: python tools/bench_plain.py [--width 1e-20] [--kinds f32,f64,2x32] [--iters 50000] [--size 3840x2160]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_PO, inputs  # noqa: E402
from test_plain_oracle import shallow_view  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", default="1e-20")
    ap.add_argument("--kinds", default="f32,f64,2x32")
    ap.add_argument("--iters", type=int, default=50000)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    W, H = [int(x) for x in a.size.split("x")]
    v = shallow_view(a.width, n_iter=a.iters, W=W, H=H)
    r = GPURenderer(0)
    assert r.InitializeMemory(W, H, 1, None, 0, 0, 0, False) == 0
    for kind in a.kinds.split(","):
        pin = inputs.PlainInputs(v, kind, host_threads=16)
        assert r.InitializePerturbPlain(0, pin) == 0
        for mode, name in ((LAV2_FULL, "full"), (LAV2_PO, "po")):
            r.enable_step_count(True)
            assert r.RenderPerturbLAv2Plain(pin, a.iters, Mode=mode) == 0
            assert r.SyncComputeStream() == 0
            st = r.read_step_count()
            r.enable_step_count(False)
            ms = []
            for _ in range(a.steps):
                assert r.RenderPerturbLAv2Plain(pin, a.iters, Mode=mode) == 0
                assert r.SyncComputeStream() == 0
                ms.append(r.last_kernel_ms())
            t = min(ms)
            print(json.dumps({"kind": kind, "mode": name, "width": a.width, "orbit": pin.count, "las": pin.la_count,
                              "ms": round(t, 3), "mpix_s": round(W * H / t / 1e3, 1),
                              "perturb_steps": st["perturb_steps"], "la_steps": st["la_steps"],
                              "at_iterations": st["at_iterations"],
                              "gsteps_s": round(st["perturb_steps"] / t / 1e6, 1),
                              "lane_util": round(st["perturb_steps"] / max(1, st["lane_slots"]), 3)}), flush=True)


if __name__ == "__main__":
    main()
