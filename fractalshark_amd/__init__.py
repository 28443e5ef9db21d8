"""fractalshark_amd -- MI355X (gfx950) per-pixel perturbation renderer for FractalShark.

The product is csrc/libfsmi355.so (hand-written HIP kernels behind the C ABI of include/fsmi355.h).  The Python
modules here are plumbing: `renderer.GPURenderer` mirrors the reference's GPURenderer over ctypes, `inputs`
builds views / reference orbits / LA / BLA tables with GMP, `tiling` row-tiles a frame over the GPUs of one node.
"""
from . import inputs  # noqa: F401
from .renderer import (GPURendererGroup, GPURenderer, LAV2_FULL, LAV2_LAO, LAV2_PO, PARITY_CPU, PARITY_CPU_GPUSTAGE,  # noqa: F401
                       T_2X32, T_2X64, T_4X32, T_4X64, T_F32, T_F64, T_HDR2X32, T_HDR32, T_HDR64)

__all__ = ["GPURenderer", "inputs", "LAV2_FULL", "LAV2_PO", "LAV2_LAO", "PARITY_CPU", "PARITY_CPU_GPUSTAGE",
           "T_F64", "T_HDR32", "T_HDR64", "T_HDR2X32", "T_F32", "T_2X32", "T_2X64", "T_4X32", "T_4X64"]
