"""CPU-only, world_size 2 over gloo: the multi-GPU row-band path (fractalshark_amd/tiling.py) -- ownership,
padding to equal slices, all-gather, device-side reassembly index -- with the CPU oracle standing in for the
kernel (the banding logic is what is under test here; the kernel side of banding is covered on the GPU by
test_gpu_parity.py::test_row_bands_reassemble_full_frame)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fractalshark_amd import tiling  # noqa: E402


def test_ownership_partitions_rows():
    for height in (1, 7, 8, 36, 180, 2160):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(height, np.int32)
            for r in range(world):
                for a, b in tiling.owned_row_ranges(height, r, world):
                    seen[a:b] += 1
                assert tiling.local_rows(height, r, world) <= tiling.max_local_rows(height, world)
            assert (seen == 1).all()
            idx = tiling.reassemble_index(height, world)
            assert len(set(idx.tolist())) == height


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    import _oracle
    from fractalshark_amd import inputs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = inputs.View.builtin(5, 64, 36)
        ob = inputs.Orbit(v)
        la = inputs.LATable(ob)
        H, rw = 36, 64
        band = tiling.band_height(1)
        max_rows = tiling.max_local_rows(H, world, band)
        local = np.zeros((max_rows, rw), np.uint32)
        k = 0
        for a, b in tiling.owned_row_ranges(H, rank, world, band):
            part = _oracle.lav2_hdr32(v, ob, la, rows=(a, b), threads=1, stage_test=1)
            local[k:k + (b - a)] = part[a:b]
            k += b - a
        lt = torch.from_numpy(local.view(np.int32))
        gathered = torch.empty((world * max_rows, rw), dtype=torch.int32)
        dist.all_gather_into_tensor(gathered, lt)
        frame = gathered.index_select(0, torch.from_numpy(tiling.reassemble_index(H, world, band)))
        frame2 = tiling.reassemble(gathered.view(world, max_rows, rw), H, world, band)
        if rank == 0:
            full = _oracle.lav2_hdr32(v, ob, la, threads=2, stage_test=1)
            ok = np.array_equal(frame.numpy().view(np.uint32), full[:H]) and \
                np.array_equal(frame2.numpy().view(np.uint32), full[:H])
            q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_reassembles_frame(native_libs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
