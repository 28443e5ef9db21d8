#!/usr/bin/env python3
"""What fs_group's delivery of a frame (gather, row order, reduction, D2H) costs that is NOT hidden behind the next frame's
kernels, measured on one GPU: N members on device 0 (peer-copy transport), View 5 at 3840x2160, K frames.

  loop A  render only                                 (K frames, one synchronisation at the end)
  loop B  render; RenderCurrent; WaitCurrent(1)        (the pipelined loop of include/fsmi355.h: two frames in flight)
  loop C  render; RenderCurrent; WaitCurrent(0)        (one frame at a time: nothing overlaps)

non-overlapped remainder per frame = (B - A) / K; what a frame's delivery costs on its own = (C - A) / K.
Usage: python tools/group_pipeline_probe.py [--members 8] [--frames 12]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fractalshark_amd import GPURendererGroup, LAV2_FULL, PARITY_CPU, T_HDR32, _capi, inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    a = ap.parse_args()
    v = inputs.View.builtin(5, a.width, a.height, antialiasing=1)
    o = inputs.Orbit(v)
    la = inputs.LATable(o, host_threads=16)
    co = [(float(c["m"]), int(c["e"])) for c in v.coords_perturb(o)]
    import torch  # page-locked host buffers (one DMA per frame); initialised before the library touches the device
    torch.cuda.init()
    g = GPURendererGroup([0] * a.members)
    assert g.InitializeMemory(a.width, a.height, 1) == 0
    assert g.InitializePerturb(1, o, la) == 0
    host = [torch.zeros(g.new_iter_buffer().shape, dtype=torch.int32, pin_memory=True) for _ in range(2)]
    host_np = [h.numpy().view(np.uint32) for h in host]
    red = [_capi.Reduction(), _capi.Reduction()]

    def render():
        assert g.RenderPerturbLAv2(*co, v.num_iterations, T=T_HDR32, Mode=LAV2_FULL, parity=PARITY_CPU) == 0

    def loop(kind):
        assert g.Sync() == 0
        t0 = time.perf_counter()
        for k in range(a.frames):
            render()
            if kind != "A":
                assert g.RenderCurrent(v.num_iterations, host_np[k % 2], red[k % 2]) == 0
                assert g.WaitCurrent(1 if kind == "B" else 0) == 0
        assert g.Sync() == 0
        return (time.perf_counter() - t0) / a.frames * 1e3

    for _ in range(3):  # warm: every member has recorded its tile costs
        render()
    g.Sync()
    out = {"members": a.members, "frames": a.frames, "frame": "%dx%d" % (a.width, a.height)}
    for rep in range(2):
        for kind in "ABC":
            out.setdefault("ms_per_frame_" + kind, []).append(round(loop(kind), 3))
    A, B, Cc = (min(out["ms_per_frame_" + k]) for k in "ABC")
    out["not_overlapped_ms_per_frame"] = round(B - A, 3)
    out["delivery_ms_alone"] = round(Cc - A, 3)
    out["gather_and_row_order_ms"] = round(g.gather_ms(), 3)
    out["checksum"] = int(host_np[(a.frames - 1) % 2][:a.height, :a.width].astype(np.uint64).sum())
    print(json.dumps(out), flush=True)
    g.close()


if __name__ == "__main__":
    main()
