"""Row-band tiling of one frame over the GPUs of one node.

The frame is embarrassingly parallel by rows once the (replicated) orbit / table are resident on every GPU.
Rank r owns the 8-row bands k*world + r (k = 0,1,...): fine interleaving, because slow (interior) pixels
cluster spatially and equal contiguous bands would be badly imbalanced.  Band height 8 = NB_THREADS_H, so a band
never splits a padding block, and it is a multiple of every legal antialiasing factor's row group only for
AA in {1,2,4} (AA=3 frames use band height 24).

The only data-path exchange is the gather of the iteration-buffer slices to rank 0 (gather_frame): one RCCL gather
of equal-sized (padded) slices -- a group of point-to-point sends, so each slice crosses its own xGMI link to GPU 0 once
and only rank 0 receives ((N-1)/N of the frame in total; an all-gather would deliver N times as much).  ~33 MB at
3840x2160 (latency-dominated next to the kernels), 531 MB at C4's 15360x8640.
"""
import numpy as np


def band_height(antialiasing=1):
    return 24 if antialiasing == 3 else 8


def owned_row_ranges(height, rank, world, band=8):
    """Global row ranges [(start, stop), ...] owned by `rank`, in local-buffer order."""
    out = []
    for start in range(rank * band, height, world * band):
        out.append((start, min(start + band, height)))
    return out


def local_rows(height, rank, world, band=8):
    return sum(b - a for a, b in owned_row_ranges(height, rank, world, band))


def max_local_rows(height, world, band=8):
    """Rows every rank's slice is padded to (rank 0 always owns the most), rounded up to 8."""
    n = local_rows(height, 0, world, band)
    return (n + 7) // 8 * 8


def reassemble(gathered, height, world, band=8):
    """gathered: array/tensor [world, max_local_rows, rounded_width] -> [height, rounded_width] in row order.

    Works on numpy arrays and torch tensors (pure indexing, stays on the device)."""
    first = gathered[0]
    if hasattr(first, "new_empty"):
        out = first.new_empty((height, first.shape[1]))
    else:
        out = np.empty((height, first.shape[1]), first.dtype)
    for rank in range(world):
        k = 0
        for a, b in owned_row_ranges(height, rank, world, band):
            out[a:b] = gathered[rank][k:k + (b - a)]
            k += b - a
    return out


def reassemble_index(height, world, band=8):
    """Flat gather index: out_rows = gathered.reshape(world*max_rows, W)[index] (one device-side index_select)."""
    m = max_local_rows(height, world, band)
    idx = np.empty(height, np.int64)
    for rank in range(world):
        k = 0
        for a, b in owned_row_ranges(height, rank, world, band):
            idx[a:b] = rank * m + k + np.arange(b - a)
            k += b - a
    return idx


def gather_slices(local, gathered, rank, world):
    """The collective alone: every rank's padded slice `local` [max_rows, W] to rank 0's `gathered` [world * max_rows, W]
    (needed on rank 0 only), enqueued on the caller's current stream.  bench.py's pipelined loop restores row order into
    a buffer of its own (torch.index_select(..., out=...)) so that no tensor is allocated per frame."""
    import torch.distributed as dist
    chunks = list(gathered.view(world, local.shape[0], local.shape[1]).unbind(0)) if rank == 0 else None
    dist.gather(local, gather_list=chunks, dst=0)


def gather_frame(local, gathered, frame_index, rank, world):
    """The data-path collective: every rank's padded slice `local` [max_rows, W] goes to rank 0 (torch.distributed.gather:
    with the nccl backend one ncclGroup of send / recv pairs), where `gathered` [world * max_rows, W] receives them back to
    back and one index_select with `frame_index` (reassemble_index) restores row order.  Returns the frame [height, W] on
    rank 0, None elsewhere.  `gathered` / `frame_index` are only needed on rank 0.  Works on any backend (gloo in the tests)."""
    import torch.distributed as dist
    chunks = list(gathered.view(world, local.shape[0], local.shape[1]).unbind(0)) if rank == 0 else None
    dist.gather(local, gather_list=chunks, dst=0)
    if rank != 0:
        return None
    return gathered.index_select(0, frame_index)


def band_copy_plan(height, rank, world, band=8):
    """The direct host path (round 6): the copies that take `rank`'s local buffer (its bands back to back) to their rows of a
    whole-frame host buffer, as fs_copy_bands_to_host (csrc/renderer.cpp) issues them -- in ROWS:
    {"dst_row", "dst_pitch_rows", "src_pitch_rows", "rows_per_band", "bands"} = ONE two-dimensional copy whose element is a whole
    band (bands that lie wholly inside the frame), and {"tail_dst_row", "tail_src_row", "tail_rows"} = the last band when the
    frame's edge cuts it (tail_rows = 0: none).  Pure host arithmetic (CPU tests; bench.py's ranks call the C function)."""
    first, stride = rank * band, world * band
    full = (height - band - first) // stride + 1 if first + band <= height else 0
    tail_start = first + full * stride
    tail_rows = max(0, min(tail_start + band, height) - tail_start) if tail_start < height else 0
    return {"dst_row": first, "dst_pitch_rows": stride, "src_pitch_rows": band, "rows_per_band": band, "bands": full,
            "tail_dst_row": tail_start, "tail_src_row": full * band, "tail_rows": tail_rows}


def apply_band_copy_plan(local, frame, plan):
    """numpy stand-in for the copy engine (tests): executes band_copy_plan on host arrays -- local [rows, W] -> frame [H, W]."""
    for k in range(plan["bands"]):
        d, s = plan["dst_row"] + k * plan["dst_pitch_rows"], k * plan["src_pitch_rows"]
        frame[d:d + plan["rows_per_band"]] = local[s:s + plan["rows_per_band"]]
    if plan["tail_rows"]:
        frame[plan["tail_dst_row"]:plan["tail_dst_row"] + plan["tail_rows"]] = \
            local[plan["tail_src_row"]:plan["tail_src_row"] + plan["tail_rows"]]
