#!/usr/bin/env python3
"""One C4 HDRFloat<double> frame per LAv2 mode (full / LA only / perturbation only is not separable: it starts from the LA result), first
frame of the view with the AT pass of its own (FSMI355_AT_SPLIT_COLD=1), for a rocprofv3 --pmc run: the difference of the frame kernel's
counters between the modes is what the perturbation loop costs.
  FSMI355_AT_SPLIT_COLD=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d OUT -- python3 tools/c4_phase_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fractalshark_amd import GPURenderer, LAV2_FULL, LAV2_LAO, PARITY_CPU_GPUSTAGE, T_HDR64  # noqa: E402

inp = bench.make_inputs("c4_hdr64")
r = GPURenderer(0)
assert r.InitializeMemory(inp["W"], inp["H"], inp["AA"], None, 0, 0, 0, False) == 0
assert r.InitializePerturb(1, inp["orbit"], 0, None, inp["la"]) == 0
for mode in (LAV2_FULL, LAV2_LAO, LAV2_FULL, LAV2_LAO):
    r.forget_tile_costs()
    assert r.RenderPerturbLAv2(None, None, None, *inp["coords"], inp["n_iter"], T=T_HDR64, Mode=mode, parity=PARITY_CPU_GPUSTAGE) == 0
    assert r.SyncComputeStream() == 0
    print("mode", mode, "kernel ms", r.last_kernel_ms(), flush=True)
r.close()
