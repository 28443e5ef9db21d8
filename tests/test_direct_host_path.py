"""CPU-only: the direct host path of the row-tiled frame (round 6) -- every rank copies its own bands straight to their rows of
ONE whole-frame host buffer, no gather and no re-order (fs_copy_bands_to_host in csrc/renderer.cpp; bench.py --host-path direct).
Here: the copy plan as pure arithmetic against the ownership function, and world_size 2 over gloo -- two rank PROCESSES fill one
frame in POSIX shared memory, the CPU oracle standing in for the kernel, rank 0 reads it when both ranks' counters say so (the
same shared-frame + counter protocol bench.py's ranks run on the GPU boxes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fractalshark_amd import tiling  # noqa: E402


def test_copy_plan_covers_exactly_the_owned_rows():
    for band in (8, 24):
        for height in (1, 7, 8, 9, 36, 75, 180, 544, 2160, 8640):
            for world in (1, 2, 3, 4, 5, 8):
                frame = np.full((height, 4), -1, np.int64)
                for rank in range(world):
                    ranges = tiling.owned_row_ranges(height, rank, world, band)
                    n = sum(b - a for a, b in ranges)
                    local = np.zeros((max(n, 1), 4), np.int64)
                    k = 0
                    for a, b in ranges:  # the local buffer: owned rows back to back, each row holds its global number
                        local[k:k + (b - a)] = np.arange(a, b)[:, None]
                        k += b - a
                    plan = tiling.band_copy_plan(height, rank, world, band)
                    assert plan["bands"] * plan["rows_per_band"] + plan["tail_rows"] == n
                    assert plan["tail_rows"] < band
                    tiling.apply_band_copy_plan(local, frame, plan)
                assert (frame == np.arange(height)[:, None]).all(), (band, height, world)


def test_copy_plan_matches_the_c_planner(native_libs):
    """fs_group_plan (csrc/group.cpp) and the Python plan agree on what a rank owns."""
    import ctypes as C

    from fractalshark_amd import _capi
    lib = _capi.render_lib()
    for height, world, band in ((36, 2, 8), (75, 3, 8), (2160, 8, 8), (8640, 8, 8), (180, 2, 24)):
        for rank in range(world):
            lr = C.c_uint32(0)
            lib.fs_group_plan(height, world, band, rank, C.byref(lr), None, None)
            plan = tiling.band_copy_plan(height, rank, world, band)
            assert plan["bands"] * band + plan["tail_rows"] == lr.value


def _worker(rank, world, port, path, q):
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
        H, rw, band = 36, 64, tiling.band_height(1)
        mm = np.memmap(path, dtype=np.uint8, mode="r+")
        frame = mm[:H * rw * 4].view(np.uint32).reshape(H, rw)
        flags = mm[H * rw * 4:H * rw * 4 + 64].view(np.int64)
        local = np.zeros((tiling.max_local_rows(H, world, band), rw), np.uint32)
        k = 0
        for a, b in tiling.owned_row_ranges(H, rank, world, band):
            part = _oracle.lav2_hdr32(v, ob, la, rows=(a, b), threads=1, stage_test=1)
            local[k:k + (b - a)] = part[a:b]
            k += b - a
        for seq in (1, 2):  # two frames: the counters only ever grow
            tiling.apply_band_copy_plan(local, frame, tiling.band_copy_plan(H, rank, world, band))
            flags[rank] = seq
            if rank == 0:
                while int(flags[:world].min()) < seq:
                    pass
                full = _oracle.lav2_hdr32(v, ob, la, threads=2, stage_test=1)
                q.put(bool(np.array_equal(frame, full[:H])))
            dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_processes_fill_one_shared_frame(native_libs, tmp_path):
    import torch.multiprocessing as mp
    path = "/dev/shm/fsmi355_test_%d" % os.getpid() if os.path.isdir("/dev/shm") else str(tmp_path / "frame")
    with open(path, "wb") as f:
        f.truncate(36 * 64 * 4 + 64)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() % 2000)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert q.get(timeout=5) is True and q.get(timeout=5) is True
    finally:
        os.unlink(path)
