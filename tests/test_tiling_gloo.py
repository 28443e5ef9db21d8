"""CPU-only, world_size 2 over gloo: the multi-GPU row-band path (fractalshark_amd/tiling.py) -- ownership,
padding to equal slices, the gather to rank 0, device-side reassembly index -- with the CPU oracle standing in for the
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
        # the collective bench.py uses: gather to rank 0 (only rank 0 holds the receive buffer and the row index)
        gathered = torch.empty((world * max_rows, rw), dtype=torch.int32) if rank == 0 else None
        index = torch.from_numpy(tiling.reassemble_index(H, world, band)) if rank == 0 else None
        frame = tiling.gather_frame(lt, gathered, index, rank, world)
        assert (frame is None) == (rank != 0)
        # ... and the form bench.py's pipelined loop uses: the collective alone into a buffer that is reused from frame to
        # frame (gather_slices), row order restored with index_select(out=...) into another one
        gathered2 = torch.full((world * max_rows, rw), -1, dtype=torch.int32) if rank == 0 else None
        tiling.gather_slices(lt, gathered2, rank, world)
        if rank == 0:
            frame2 = tiling.reassemble(gathered.view(world, max_rows, rw), H, world, band)
            frame3 = torch.empty((H, rw), dtype=torch.int32)
            torch.index_select(gathered2, 0, index, out=frame3)
            full = _oracle.lav2_hdr32(v, ob, la, threads=2, stage_test=1)
            ok = np.array_equal(frame.numpy().view(np.uint32), full[:H]) and \
                np.array_equal(frame2.numpy().view(np.uint32), full[:H]) and \
                np.array_equal(frame3.numpy().view(np.uint32), full[:H])
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


# ---- bench.py's own rank launcher (python bench.py --gpus N without torchrun)
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fs_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)  # top level of bench.py imports the standard library only
    return m


def test_bench_builds_its_own_launch_command(monkeypatch):
    b = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert b.rank_launch_command(1, ["--gpus", "1"]) is None
    cmd = b.rank_launch_command(8, ["--gpus", "8", "--steps", "3", "--workload", "c5_bla"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--workload", "c5_bla"] and cmd[-7].endswith("bench.py")
    # under a launcher (the driver's torchrun command) the process is a rank: nothing is started
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert b.rank_launch_command(8, ["--gpus", "8"]) is None


_RANK_SCRIPT = r'''
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
print("banner from rank", dist.get_rank(), file=sys.stderr)
if "--fail" in sys.argv and dist.get_rank() == 1:
    sys.exit(7)
if dist.get_rank() == 0:
    print("library noise on stdout")
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:]}))
dist.destroy_process_group()
'''


def test_bench_launcher_relays_the_json_line_and_the_exit_code(tmp_path, monkeypatch, capfd):
    """launch_ranks() with a stand-in rank script (gloo, 2 ranks): ONE JSON line on stdout, child's code returned."""
    import json
    b = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    script = tmp_path / "rank_script.py"
    script.write_text(_RANK_SCRIPT)
    rc = b.launch_ranks(b.rank_launch_command(2, ["--gpus", "2"], script=str(script)))
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--gpus", "2"]
    rc = b.launch_ranks(b.rank_launch_command(2, ["--gpus", "2", "--fail"], script=str(script)))
    assert rc != 0
