#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpix/s of the iteration buffer, View #5 at 3840x2160, HDRx32 LAv2.

  python bench.py --gpus N --steps K --warmup W [--workload c3_lav2|c1_direct|c2_po|c5_bla|c4_hdr64|c4_2x32|c4_scaled]
                  [--host-path gather|direct]

N > 1 is one rank per GPU under torch.distributed (backend "nccl" = RCCL).  When bench.py is started WITHOUT a launcher
(no WORLD_SIZE / RANK in the environment) and --gpus N > 1, it starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>` itself, as a CHILD process, before torch is imported
or any HIP call is made, relays the child's JSON line and exits with its code.  Started under a launcher (the driver's
torchrun command) it is a rank.  Every workload runs at every N.

One "step" = one full frame in the window SURVEY.md section 8(d) defines (the reference's m_PerPixel timer,
Fractal.cpp:2842 -> :1539): launch of the iteration kernel -> iteration buffer resident in HOST memory.  For N > 1
each rank renders its interleaved 8-row bands, the slices are gathered to rank 0 with one RCCL gather over xGMI
(fractalshark_amd/tiling.py: grouped send / recv, only rank 0 receives), rank 0 restores row order on the device and copies
the frame to the host.  Inputs (reference orbit, LA
table) are generated on the host with GMP *before* the timed region and are resident in HBM when it starts;
`value` = W*H*K / t with t = max over ranks of the barrier-bracketed wall time.  Prints ONE JSON line on rank 0.

Extra objects (round contract):
  roofline      dominant kernel of the workload.  The perturbation kernels are VALU-bound (scalar complex arithmetic,
                no dense contraction; orbit + table are L2 resident), so `bound` is "valu": achieved = executed
                pixel-steps x 18 FP32 flop / average kernel duration measured with HIP
                events on the renderer's compute stream; peak = 157.3 TFLOP/s FP32 vector (MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (restatement of the reference's multithreaded CPU RenderAlgorithm, same row-claiming
                thread pool) timed on this host's cores over a bounded sample of rows of the same frame -- the
                UNMODIFIED function; `cpu_baseline_patched` is the same with the LA stage test in the GPU direction
                (SURVEY.md section 0.1 / 8(d) asks for both).
  secondary     (c3_lav2 only) the same frame with FS_PARITY_CPU_GPUSTAGE, timed the same way: the configuration in
                which the LA stages are actually used (the literal CPU stage test skips every stage at View 5).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_STEP = 18            # complex dz*(2Z+dz)+dc, |Z+dz|^2, |dz|^2
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_FP64_VECTOR_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
FLOP_PER_BLA_JUMP = 22
FLOP_PER_DIRECT_ITERATION = 8  # z*z + c in binary64 with |z|^2: 4 multiplications + 4 additions (Fractal.cpp:2148-2183)
TRAFFIC_FILE = "r06_traffic.json"  # profiles/: PMC passes of THIS round's kernels (tools/profile_round6.sh + _collect.py)

WORKLOADS = {  # name: (view, width, height, tag in config.workload, dominant kernel)
    "c3_lav2": (5, 3840, 2160, "hdrx32_lav2_full", "k_lav2_hdr32_fast"),
    "c1_direct": (0, 1024, 768, "f64_direct", "k_direct_f64"),
    "c2_po": (5, 1920, 1080, "hdrx32_po", "k_perturb_scalar<float, false>"),
    "c5_bla": (19, 7680, 4320, "hdrx32_bla", "k_bla_hdr32_fast"),
    "c4_hdr64": (14, 3840, 2160, "hdrx64_lav2_full_aa4", "k_lav2_hdr64"),
    "c4_2x32": (14, 3840, 2160, "hdrx2x32_lav2_full_aa4", "k_lav2_2x32"),
    "c4_scaled": (14, 3840, 2160, "hdrx32_scaled_aa1_itercap", "k_scaled_hdr32_fast"),
}


def effective_cpus():
    """Usable host threads: the scheduler affinity capped by the cgroup CPU quota (the GPU boxes expose 256
    logical CPUs but cap the container at a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    """Model string of the host CPU (/proc/cpuinfo) and the logical CPUs the OS reports -- the CPU baseline is only
    interpretable next to them (boxes of the pool differ, and the container is capped well below the logical count)."""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def oracle_frame_record(workload_key, parity, n_iterations):
    """tests/golden/frame_crcs.json: the WHOLE frame of this workload as the CPU oracle renders it (tests/golden/make_frame_crcs.py,
    build container, no GPU involved) -- {"crc32", "sum", "band_rows", "band_crc32", ...} or None when no oracle frame is committed
    for this workload / parity / cap."""
    try:
        table = json.load(open(os.path.join(ROOT, "tests", "golden", "frame_crcs.json")))
    except (OSError, ValueError):
        return None
    rec = table.get("%s|%s|%d" % (workload_key, parity, n_iterations))
    return rec if isinstance(rec, dict) and rec.get("source") == "oracle" else None


def frame_crc32(frame, H, W):
    """zlib CRC-32 of rows 0..H-1 x columns 0..W-1 of the iteration buffer, uint32 little-endian, row-major: the value
    tests/golden/frame_crcs.json holds for the oracle's frame."""
    import zlib

    import numpy as np
    return "%08x" % (zlib.crc32(np.ascontiguousarray(frame[:H, :W]).astype("<u4", copy=False).tobytes()) & 0xFFFFFFFF)


def make_inputs(wl, view_id=-1, width=0, height=0, iter_cap=0, parity=None):
    """Everything a workload's frame is a function of, built on the host the way main() hands it to the renderer: the view,
    the reference orbit (GMP), the LA / BLA table, the pixel -> delta-c coefficients, the iteration cap and the parity mode.
    Shared with tests/golden/make_frame_crcs.py, which renders the same frames with the CPU oracle.  No torch, no HIP.
    (The LA builder's thread count only matters through min(orbit entries / 50000, threads), LAReference.cpp:236-251: 1 for
    View 5's 16 046 entries and 2 for View 14's 116 695 on every host with two threads or more.)"""
    from fractalshark_amd import inputs
    dview, dw, dh = WORKLOADS[wl][:3]
    view_id = dview if view_id < 0 else view_id
    width = dw if width <= 0 else width
    height = dh if height <= 0 else height
    if parity is None:
        parity = "cpu_gpustage" if wl == "c4_hdr64" else "cpu"
    is_lav2 = wl in ("c3_lav2", "c4_hdr64", "c4_2x32")
    is2x32 = wl == "c4_2x32"
    is_scaled = wl == "c4_scaled"
    is64 = wl in ("c4_hdr64", "c4_2x32")  # the 2x32 inputs are derived from the HDRFloat<double> ones
    is_direct = wl == "c1_direct"
    view = inputs.View.builtin(view_id, width, height, antialiasing=None if is64 else 1)
    if is_direct:
        # BASELINE config C1: no reference orbit, no table -- CalcCpuHDR<uint32_t,double,double> iterates every pixel itself
        AA = view.antialiasing
        co = view.coords_direct_f64(AA)
        return {"workload": wl, "view_id": view_id, "width": width, "height": height, "parity": parity or "cpu", "is_lav2": False,
                "is2x32": False, "is_scaled": False, "is64": False, "is_direct": True, "view": view, "orbit": None, "la": None,
                "orbit2": None, "la2": None, "bla": None, "AA": AA, "W": view.width * AA, "H": view.height * AA,
                "n_iter": iter_cap if iter_cap > 0 else view.num_iterations, "coords_arr": co, "coords": None,
                "key": "view%d_%dx%d_%s" % (view_id, view.width * AA, view.height * AA, WORKLOADS[wl][3])}
    orbit = inputs.Orbit(view, is64=is64)
    la = inputs.LATable(orbit, host_threads=max(2, effective_cpus()), use_small_exponents=is2x32) if is_lav2 else None
    orbit2 = inputs.Orbit2x32(orbit) if is2x32 else None
    la2 = inputs.LATable2x32(la) if is2x32 else None
    bla = inputs.BLATable(orbit) if wl == "c5_bla" else None
    AA = view.antialiasing
    n_iter = view.num_iterations
    if iter_cap > 0:
        n_iter = iter_cap
    elif is_scaled:
        n_iter = 65536
    if is2x32:
        coords_arr = view.coords_perturb_2x32(orbit2)
        coords = [(float(c["head"]), float(c["tail"]), int(c["e"])) for c in coords_arr]
    else:
        coords_arr = view.coords_perturb(orbit)
        coords = [(float(c["m"]), int(c["e"])) for c in coords_arr]
    return {"workload": wl, "view_id": view_id, "width": width, "height": height, "parity": parity, "is_lav2": is_lav2,
            "is2x32": is2x32, "is_scaled": is_scaled, "is64": is64, "is_direct": False, "view": view, "orbit": orbit, "la": la, "orbit2": orbit2,
            "la2": la2, "bla": bla, "AA": AA, "W": view.width * AA, "H": view.height * AA, "n_iter": n_iter,
            "coords_arr": coords_arr, "coords": coords,
            "key": "view%d_%dx%d_%s" % (view_id, view.width * AA, view.height * AA, WORKLOADS[wl][3])}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3_lav2",
                    help="c3_lav2 (default, the headline config): View 5 3840x2160 HDRx32 LAv2 Full; "
                         "c2_po: View 5 1920x1080 HDRx32 perturbation only; c5_bla: View 19 7680x4320 HDRx32 BLA; "
                         "c4_hdr64: View 14 (zoom 2^-21645) 3840x2160 x AA4 = 15360x8640 with HDRFloat<double> LAv2 -- the "
                         "CPU-twinned form of C4 (use --parity cpu_gpustage: the literal CPU function needs ~6e5 "
                         "perturbation steps per pixel there); c4_2x32: the same frame with HDRFloat<CudaDblflt> "
                         "(GpuHDRx2x32PerturbedLAv2), checked against the restated CUDA kernel (no CPU twin exists); "
                         "c4_scaled: View 14 at 3840x2160 (AA 1) through GpuHDRx32PerturbedScaled with the iteration cap "
                         "lowered to --iter-cap (perturbation-only rendering of this view needs the full 2^31 steps for "
                         "almost every pixel), checked against the restated CUDA kernel")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--view", type=int, default=-1)
    ap.add_argument("--iter-cap", type=int, default=0,
                    help="override the view's iteration cap (default: the view's own; 65536 for c4_scaled)")
    ap.add_argument("--parity", choices=["cpu", "cpu_gpustage"], default=None,
                    help="cpu = literal reference CPU function (bit-exact vs Cpu32PerturbedBLAV2HDR); "
                         "cpu_gpustage = same arithmetic, LA stage test in the GPU/FractalZoomer direction. "
                         "Default: cpu, except c4_hdr64 (cpu_gpustage: the literal CPU direction skips every LA "
                         "stage at View 14 and iterates towards the 2^31 cap)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the frame timed on the CPU (0 = a per-workload multiple of the usable host threads, "
                         "about 10-30 s of CPU work)")
    ap.add_argument("--variant", default="",
                    help="comma-separated A/B selections of fs_set_kernel_variant: literal | noscale, lds_orbit, refill, natural_tile_order "
                         "(default: the tuned kernels)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-frame (natural tile order) latency measurement")
    ap.add_argument("--natural-tile-order", action="store_true",
                    help="FS_VARIANT_NATURAL_TILE_ORDER: every frame in natural tile order (no self-recorded longest-first)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (default) = RCCL: the slices go to rank 0 over xGMI on the device.  gloo = the same N-rank program "
                         "with a HOST-STAGED exchange (slice -> page-locked host -> gloo gather -> rank 0 puts the rows in order on "
                         "the host): the functional N > 1 path where RCCL between the ranks cannot come up -- e.g. two rank "
                         "processes sharing one device (--share-device), which is how a one-GPU box runs the whole of main() "
                         "with a peer.  Not a performance path: the exchange blocks the host")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses device 0 (tests on a one-GPU box; needs --dist-backend gloo: RCCL refuses two ranks "
                         "on one device)")
    ap.add_argument("--host-path", choices=["gather", "direct"], default="gather",
                    help="N > 1: how the frame reaches host memory.  gather (default) = the slices go to rank 0 (RCCL over xGMI), one "
                         "kernel restores row order, ONE copy over rank 0's PCIe link.  direct = every rank copies its own bands "
                         "straight to their rows of a shared, page-locked frame (POSIX shared memory registered in every rank; "
                         "fs_copy_bands_to_host: one 2-D copy per rank, destination pitch = band stride) over ITS OWN link -- no "
                         "gather, no re-order kernel, N links instead of one.  Same frame either way")
    ap.add_argument("--no-build", action="store_true",
                    help="never spawn a compiler (same as FS_NO_BUILD=1): required under rocprofv3, see tools/pmc_passes.sh")
    return ap.parse_args()


def rank_launch_command(gpus, argv, script=None, port=None):
    """The command that runs `script` (default: this file) with `argv` as `gpus` ranks of one node, or None when this
    process is a rank already (a launcher set WORLD_SIZE / RANK) or one rank is asked for.  Pure: no torch, no HIP."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    if port is None:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__), *argv]


def launch_ranks(cmd):
    """Run the rank launcher as a child and relay it: its stdout (the ONE JSON line of rank 0) goes to our stdout, its
    stderr passes through, its exit code is ours.  The parent never imports torch nor touches the GPU (replacing a process
    that has initialised the GPU is forbidden on the GPU pool; a child of a clean parent is always fine)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL / IPC between the ranks needs dmabuf IPC on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if json_lines:
        print(json_lines[-1], flush=True)
    return p.returncode if p.returncode != 0 or json_lines else 1


def main():
    args = parse()
    cmd = rank_launch_command(args.gpus, sys.argv[1:])
    if cmd is not None:
        # under a profiler the preloaded tool library has initialised the GPU in THIS process already: starting the
        # launcher from here would be the fork + exec of a GPU-initialised process that the pool forbids
        # (tools/pmc_passes.sh) -- profile one rank (--gpus 1), or start the ranks under the profiler yourself
        # (a preload by itself is not a profiler: the GPU boxes preload their own exec guard into every process)
        traced = [k for k in os.environ if k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER"))] + \
                 (["LD_PRELOAD"] if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() else [])
        if traced:
            print("bench.py: --gpus %d without a launcher under a profiler / preload (%s): refusing to start the rank "
                  "launcher from a process that may have initialised the GPU" % (args.gpus, ", ".join(sorted(traced))),
                  file=sys.stderr)
            sys.exit(2)
        sys.exit(launch_ranks(cmd))
    # stdout carries exactly ONE JSON line.  Libraries write banners there (RCCL prints its version block to stdout when the
    # first communicator comes up), so for the whole run file descriptor 1 is pointed at stderr and the real stdout is kept
    # for the result line only.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # RCCL / IPC between the ranks needs dmabuf IPC on these hosts (hipIpcGetMemHandle: invalid argument without it).  Set in
    # EVERY rank before the runtime comes up -- a rank started by somebody else's torchrun never passes through launch_ranks
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch

    from fractalshark_amd import (GPURenderer, LAV2_FULL, PARITY_CPU, PARITY_CPU_GPUSTAGE, T_F64, T_HDR2X32, T_HDR32, T_HDR64,
                                  _build, inputs, tiling)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FS_FORCE_DIST=1 exercises the RCCL gather path with a single rank (1-GPU boxes)
    distributed = world > 1 or os.environ.get("FS_FORCE_DIST") == "1"
    no_build = args.no_build or os.environ.get("FS_NO_BUILD") == "1"
    if no_build:
        os.environ["FS_NO_BUILD"] = "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # FS_FORCE_DIST=1 without a launcher: a one-rank job
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_device:
            assert args.dist_backend == "gloo", "--share-device needs --dist-backend gloo (RCCL refuses two ranks on one device)"
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)
    host_staged = distributed and args.dist_backend == "gloo"
    red_device = "cpu" if host_staged else "cuda"  # where the few scalars that cross ranks live
    if rank == 0:
        if no_build:
            # under a profiler (rocprofv3 preloads a library that has initialised the GPU) no compiler may be spawned:
            # a stale or missing library is an error, not a rebuild
            if not _build.up_to_date():
                raise RuntimeError("native libraries are missing or stale and --no-build / FS_NO_BUILD=1 forbids "
                                   "compiling: run `python -c 'from fractalshark_amd import _build; _build.build_all()'` first")
        else:
            _build.build_all()
    if distributed:
        dist.barrier()

    if GPURenderer.TestCudaIsWorking() == 0:
        raise RuntimeError("no usable HIP device; there is no CPU fallback")

    # ---- inputs (host, outside the timed region)
    wl = args.workload
    wl_tag, wl_kernel = WORKLOADS[wl][3], WORKLOADS[wl][4]
    t0 = time.time()
    inp = make_inputs(wl, args.view, args.width, args.height, args.iter_cap, args.parity)
    t_inputs = time.time() - t0
    args.view, args.width, args.height, args.parity = inp["view_id"], inp["width"], inp["height"], inp["parity"]
    is_lav2, is2x32, is_scaled, is64 = inp["is_lav2"], inp["is2x32"], inp["is_scaled"], inp["is64"]
    is_direct = inp["is_direct"]
    inp_default_parity = "cpu_gpustage" if wl == "c4_hdr64" else "cpu"  # (the parity the profiles were taken in)
    view, orbit, la, orbit2, la2, bla = inp["view"], inp["orbit"], inp["la"], inp["orbit2"], inp["la2"], inp["bla"]
    AA, W, H, n_iter = inp["AA"], inp["W"], inp["H"], inp["n_iter"]
    coords_arr, coords = inp["coords_arr"], inp["coords"]
    parity = PARITY_CPU if args.parity == "cpu" else PARITY_CPU_GPUSTAGE

    r = GPURenderer(local_rank)
    err = r.InitializeMemory(W, H, AA, None, 0, 0, 0, False)
    assert err == 0, GPURenderer.ConvertErrorToString(err)
    T_TAG = T_HDR2X32 if is2x32 else (T_HDR64 if is64 else T_HDR32)
    vsel = [x for x in args.variant.split(",") if x]
    if vsel:
        unknown = set(vsel) - {"literal", "noscale", "lds_orbit", "refill", "natural_tile_order"}
        assert not unknown, "unknown --variant entries: %s" % sorted(unknown)
        e = r.set_kernel_variant(1 if "literal" in vsel else (2 if "noscale" in vsel else 0),
                                 lds_orbit="lds_orbit" in vsel, refill="refill" in vsel,
                                 natural_tile_order="natural_tile_order" in vsel)
        assert e == 0, GPURenderer.ConvertErrorToString(e)
    lib = r._lib
    if is2x32:
        assert r.InitializePerturb(1, orbit2, 0, None, la2) == 0
    elif is_lav2:
        assert r.InitializePerturb(1, orbit, 0, None, la) == 0
    elif is_direct:
        pass  # (no inputs beyond the coordinates)
    elif is_scaled:
        # the reference re-uploads both PerturbExtras::Bad orbits inside every RenderPerturbBLAScaled call
        # (GPU_Render.cu:1324-1345); here once, outside the timed region (inputs resident in HBM)
        assert lib.fs_upload_orbit_scaled(r._h, T_HDR32, 4, orbit.bad_data_ptr, orbit.bad_f32_data_ptr, orbit.count,
                                          orbit.period) == 0
    else:
        # the reference re-uploads orbit + BLA table inside every RenderPerturbBLA call (GPU_Render.cu:1464-1479);
        # here they are uploaded once, outside the timed region (inputs resident in HBM)
        assert lib.fs_upload_orbit(r._h, 1, T_HDR32, 4, orbit.data_ptr, orbit.count, orbit.count, orbit.period) == 0
        if bla is not None:
            assert lib.fs_upload_bla(r._h, T_HDR32, bla.level_ptrs, bla.level_sizes, bla.num_levels, bla.lm2) == 0
        else:
            assert lib.fs_upload_bla(r._h, T_HDR32, None, None, 0, 0) == 0
    # informational: the same LA table built on the device (fs_build_la) instead of the host builder
    la_device_ms = None
    if is_lav2 and not is2x32 and orbit.count < 100000:  # below the reference's multi-threading threshold the tables agree
        t1 = time.perf_counter()
        if r.BuildLAOnDevice(orbit) == 0:
            la_device_ms = round((time.perf_counter() - t1) * 1e3, 3)
        assert r.InitializePerturb(2, orbit, 0, None, la) == 0  # the timed frames use the uploaded host table
    band = tiling.band_height(AA)
    rw = r.rounded_width
    rows_padded = (H + 7) // 8 * 8
    # ---- buffers of the frame pipeline.  TWO of everything a frame passes through, in rotation, so that
    # frame k+1's kernel runs while frame k is gathered, put back in row order and copied to the host:
    #   local[b]     the rank's iteration buffer (caller-owned: fs_set_external_iter_buffer), written by the kernel
    #   gathered[b]  rank 0, N > 1: the ranks' padded slices back to back (one RCCL gather)
    #   frame_dev[b] rank 0, N > 1: the frame in row order (one index_select into a fixed buffer)
    #   host[b]      rank 0: the frame in page-locked host memory (the reference's ItersMemoryContainer) -- one DMA
    # Streams: the renderer's compute stream (kernel), `post` (gather + row order), `copy` (D2H); events order them and
    # the HOST waits on the D2H event of a frame, never on the device.
    NB = 2
    if distributed:
        assert r.SetRowBands(rank * band, band, world * band) == 0
        local_rows = tiling.max_local_rows(H, world, band)
    else:
        local_rows = rows_padded
    local = [torch.zeros((local_rows, rw), dtype=torch.int32, device="cuda") for _ in range(NB)]
    local_bytes = local[0].numel() * local[0].element_size()
    direct_path = distributed and args.host_path == "direct"
    shm_files = shm_maps = flags = None
    host_ptr = None
    if direct_path:
        # The frame lives in POSIX shared memory that EVERY rank maps and page-locks (fs_host_register = hipHostRegister, portable):
        # rank r's copy engine writes rank r's bands into it over rank r's own PCIe link.  One small shared array of frame
        # counters tells rank 0 (the consumer) when every rank's bands of a frame have landed.
        names = [None]
        if rank == 0:
            names = ["/dev/shm/fsmi355_bench_%d" % os.getpid()]
            with open(names[0], "wb") as f:
                f.truncate(NB * rows_padded * rw * 4 + 4096)
        dist.broadcast_object_list(names, src=0)
        shm_files = names
        mm = np.memmap(names[0], dtype=np.uint8, mode="r+")
        shm_maps = mm
        assert lib.fs_host_register(mm.ctypes.data, mm.nbytes) == 0, "hipHostRegister of the shared frame failed"
        frame_bytes = rows_padded * rw * 4
        host_np = [mm[b * frame_bytes:(b + 1) * frame_bytes].view(np.uint32).reshape(rows_padded, rw) for b in range(NB)]
        host_ptr = [mm.ctypes.data + b * frame_bytes for b in range(NB)]
        flags = mm[NB * frame_bytes:NB * frame_bytes + 4096].view(np.int64)  # flags[r] = frames of rank r that are in the host frame
        host_t = None
        dist.barrier()
    else:
        host_t = [torch.zeros((rows_padded, rw), dtype=torch.int32, pin_memory=True) for _ in range(NB)] if rank == 0 else None
        host_np = [t.numpy().view(np.uint32) for t in host_t] if rank == 0 else None
    gathered = frame_dev = frame_index = stage = None
    if direct_path:
        pass  # (no gather buffers at all)
    elif host_staged:
        stage = torch.zeros((local_rows, rw), dtype=torch.int32, pin_memory=True)
        if rank == 0:
            gathered = [torch.empty((world * local_rows, rw), dtype=torch.int32)]
            frame_index = torch.from_numpy(tiling.reassemble_index(H, world, band))
    elif distributed and rank == 0:
        gathered = [torch.empty((world * local_rows, rw), dtype=torch.int32, device="cuda") for _ in range(NB)]
        frame_dev = [torch.empty((H, rw), dtype=torch.int32, device="cuda") for _ in range(NB)]
        frame_index = torch.from_numpy(tiling.reassemble_index(H, world, band)).cuda()
    render_stream = torch.cuda.ExternalStream(r.compute_stream)
    post_stream = torch.cuda.Stream()
    copy_stream = torch.cuda.Stream()
    ev_render = [torch.cuda.Event() for _ in range(NB)]
    ev_consumed = [torch.cuda.Event() for _ in range(NB)]  # local[b] has been read by everything that reads it
    ev_host = [torch.cuda.Event() for _ in range(NB)]      # host[b] holds the frame
    used = [False] * NB
    state = {"k": 0, "last": None}
    seq_of = [0] * NB  # direct path: the frame number (1, 2, ...) buffer b holds

    def enqueue_frame(par):
        """Launch one frame and everything behind it (asynchronous).  Returns the buffer index it lands in."""
        b = state["k"] % NB
        state["k"] += 1
        if used[b]:
            render_stream.wait_event(ev_consumed[b])  # the kernel must not overwrite local[b] before it has been read
        assert r.SetExternalIterBuffer(local[b].data_ptr(), local_bytes) == 0
        if is_lav2:
            e = r.RenderPerturbLAv2(None, None, None, *coords, n_iter, T=T_TAG, Mode=LAV2_FULL, parity=par)
        elif is_direct:
            e = lib.fs_render_direct(r._h, T_F64, coords_arr.ctypes.data, n_iter)
        elif is_scaled:
            e = lib.fs_render_scaled(r._h, T_HDR32, coords_arr.ctypes.data, n_iter)
        else:
            e = lib.fs_render_bla(r._h, T_HDR32, coords_arr.ctypes.data, n_iter)
        assert e == 0, GPURenderer.ConvertErrorToString(e)
        ev_render[b].record(render_stream)
        src = local[b]
        if direct_path:
            # every rank: its own bands -> their rows of the shared frame, on its copy stream behind its kernel
            copy_stream.wait_event(ev_render[b])
            e = r.CopyBandsToHost(host_ptr[b], local[b].data_ptr(), copy_stream.cuda_stream)
            assert e == 0, GPURenderer.ConvertErrorToString(e)
            ev_host[b].record(copy_stream)
            ev_consumed[b].record(copy_stream)
            used[b] = True
            state["last"] = b
            seq_of[b] = state["k"]
            return b
        if host_staged:
            # the same exchange through host memory: wait for the kernel, slice -> page-locked host, gloo gather, rank 0 puts
            # the rows in order straight into the frame's host buffer.  Blocking: frames do not overlap in this mode
            ev_render[b].synchronize()
            stage.copy_(local[b])
            tiling.gather_slices(stage, gathered[0] if rank == 0 else None, rank, world)
            if rank == 0:
                torch.index_select(gathered[0], 0, frame_index, out=host_t[b][:H])
            ev_consumed[b].record(render_stream)
            ev_host[b].record(render_stream)
            used[b] = True
            state["last"] = b
            return b
        if distributed:
            with torch.cuda.stream(post_stream):
                post_stream.wait_event(ev_render[b])  # device-side: the gather waits for the kernel, not the host
                if rank == 0 and used[b]:
                    post_stream.wait_event(ev_host[b])  # gathered[b] / frame_dev[b] are still being copied out (frame k - NB)
                tiling.gather_slices(local[b], gathered[b] if rank == 0 else None, rank, world)
                ev_consumed[b].record(post_stream)
                if rank == 0:
                    torch.index_select(gathered[b], 0, frame_index, out=frame_dev[b])
                    src = frame_dev[b]
        if rank == 0:
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev_render[b])
                if distributed:
                    copy_stream.wait_stream(post_stream)
                host_t[b][:src.shape[0]].copy_(src, non_blocking=True)
                ev_host[b].record(copy_stream)
                if not distributed:
                    ev_consumed[b].record(copy_stream)
        elif not distributed:
            ev_consumed[b].record(render_stream)
        used[b] = True
        state["last"] = b
        return b

    def wait_frame(b):
        """Host waits until frame in buffer b is where it has to be (rank 0: in host memory)."""
        if direct_path:
            ev_host[b].synchronize()      # my bands are in the shared frame ...
            flags[rank] = seq_of[b]
            if rank == 0:                 # ... and the consumer waits for everybody's
                while int(flags[:world].min()) < seq_of[b]:
                    pass
            return
        (ev_host[b] if rank == 0 else ev_consumed[b]).synchronize()

    def one_frame(par):
        """One frame, launch -> iteration buffer in host memory (rank 0), synchronously: the frame LATENCY."""
        b = enqueue_frame(par)
        wait_frame(b)
        return b

    def barrier_sync():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(par, cold=False):
        """EXACTLY args.steps frames between barrier + synchronize on both sides, pipelined: while frame k's slices are
        gathered and copied out, frame k+1's kernel runs; the host waits for frame k-1's host buffer before it launches
        frame k+1.  cold: every frame is the FIRST frame of its view (whatever the renderer recorded is dropped before each
        launch -- a viewer that zooms changes the coordinates with every frame).
        -> (seconds, max over ranks; kernel ms of this rank's frames)"""
        barrier_sync()
        t0 = time.perf_counter()
        prev = None
        for _ in range(args.steps):
            if cold:
                r.forget_tile_costs()
            b = enqueue_frame(par)
            if prev is not None:
                wait_frame(prev)
            prev = b
        wait_frame(prev)
        barrier_sync()
        el = max_over_ranks(time.perf_counter() - t0)
        return el, r.kernel_ms_history(min(args.steps, 64))

    def latency(par, frames, cold=False):
        """`frames` frames one at a time (nothing overlaps): mean wall ms launch -> host, max over ranks; kernel ms."""
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            if cold:
                r.forget_tile_costs()
            one_frame(par)
        barrier_sync()
        el = max_over_ranks(time.perf_counter() - t0)
        return el / frames * 1e3, r.kernel_ms_history(min(frames, 64))

    def count_steps(par):
        # Executed work per frame is a pure function of the inputs: count it once in an untimed launch with the
        # instrumented kernel build (wave-reduced atomics); the timed launches run the uninstrumented kernel.
        r.enable_step_count(True)
        one_frame(par)
        st = r.read_step_count()
        r.enable_step_count(False)
        if distributed:
            keys = ("perturb_steps", "at_iterations", "la_steps")
            t = torch.tensor([st[k] for k in keys], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            st = dict(st)
            st.update({k: float(x) for k, x in zip(keys, t.tolist())})
        return st

    tile_order_on = "natural_tile_order" not in vsel and not args.natural_tile_order
    if not tile_order_on:
        r.set_kernel_variant(1 if "literal" in vsel else (2 if "noscale" in vsel else 0), lds_orbit="lds_orbit" in vsel,
                             refill="refill" in vsel, natural_tile_order=True)
    st = count_steps(parity)
    for _ in range(args.warmup):
        one_frame(parity)
    # ... up to the view's steady state: the pixel order of the HDRFloat<double> / <CudaDblflt> frames is made for a view that is
    # rendered a second time (its first frame runs as it is, its second records and sorts, the third and later ones run ordered)
    if tile_order_on and wl in ("c4_hdr64", "c4_2x32"):
        for _ in range(3):
            if r.last_frame_tile_ordered():
                break
            one_frame(parity)
    # the frame the steady state starts from: a frame in natural order has run (count_steps), its recorded costs order the
    # frames that follow ("warm": every timed frame, like every frame after the first of a real sequence of frames)
    # HDRFloat<double> / <CudaDblflt> (advisor, round 5): their pixel order is keyed by the view's COORDINATES, so it only exists
    # for a view that is rendered three times; a viewer that zooms never sees it.  `value` of these two workloads is therefore
    # the COLD figure -- every timed frame a first frame -- and the warm one is reported beside it (value_warm).  The
    # HDRFloat<float> tile costs are keyed without the coordinates and do carry over from frame to frame of a zoom: warm.
    cold_headline = wl in ("c4_hdr64", "c4_2x32") and tile_order_on
    elapsed_warm = kernel_ms_warm = None
    if cold_headline:
        elapsed_warm, kernel_ms_warm = timed(parity)
        warm_split = r.kernel_ms_split_history(min(args.steps, 64))
        elapsed, kernel_ms = timed(parity, cold=True)
        kernel_split = r.kernel_ms_split_history(min(args.steps, 64))
        for _ in range(3):  # (leave the renderer warm again for the latency figures below)
            if r.last_frame_tile_ordered():
                break
            one_frame(parity)
    else:
        elapsed, kernel_ms = timed(parity)
        kernel_split = r.kernel_ms_split_history(min(args.steps, 64))
        warm_split = None
    avg_kernel_ms = max_over_ranks(sum(kernel_ms) / len(kernel_ms))
    is_po = wl == "c2_po"
    ordered_frames = bool(r.last_frame_tile_ordered()) if (is_lav2 or is_po) else False
    if cold_headline:
        ordered_frames = True  # (the warm frames above were; the latency figures below measure both)
    perturb_steps, at_iters, la_steps = float(st["perturb_steps"]), float(st["at_iterations"]), float(st["la_steps"])
    # HDRFloat<double> / <CudaDblflt> LAv2: the AT iterations the launch actually RAN (statistics word 5 of k_lav2_lit / k_lav2_2x32: the AT loop's cycle search
    # spares a pixel inside the set most of the iterations the limit asks of it -- same state, same count, same frame).  The
    # roofline prices executed work; at_iterations_per_launch stays what the reference's loop would run.
    at_iters_executed = float(st.get("careful_steps", 0)) if (wl in ("c4_hdr64", "c4_2x32") and st.get("careful_steps")) else at_iters
    if wl == "c4_hdr64" and tile_order_on and not cold_headline and st.get("scaled_steps"):
        # the timed frames run PerformAT in a pass of its own, whose lanes are grouped by what each pixel needs by itself
        # (statistics word 6 of the counting launch, which iterates inside the frame's kernel and waits for the wave's slowest)
        at_iters_executed = float(st["scaled_steps"])
    lane_slots = st["lane_slots"]
    b_main = state["last"]
    checksum = int(host_np[b_main][:H, :W].astype(np.uint64).sum()) if rank == 0 else None
    frame_main = host_np[b_main].copy() if rank == 0 else None
    crc_main = frame_crc32(frame_main, H, W) if rank == 0 else None
    # latency of ONE frame (launch -> host, nothing overlapped), warm and cold (natural tile order: what the FIRST frame of a
    # view costs; fs_forget_tile_costs before every frame)
    n_lat = max(1, min(args.steps, 5))
    lat_warm_ms, lat_warm_k = latency(parity, n_lat)
    lat_cold_ms = lat_cold_k = None
    if ordered_frames and not args.no_cold:
        lat_cold_ms, lat_cold_k = latency(parity, n_lat, cold=True)
        one_frame(parity)  # (leave the renderer warm again)
    frame_timing = None
    if rank == 0:
        frame_timing = {
            "sustained_ms_per_frame": round(elapsed / args.steps * 1e3, 3),
            "latency_ms_warm": round(lat_warm_ms, 3), "latency_kernel_ms_warm": round(sum(lat_warm_k) / len(lat_warm_k), 3),
            "latency_ms_cold": round(lat_cold_ms, 3) if lat_cold_ms is not None else None,
            "latency_kernel_ms_cold": round(sum(lat_cold_k) / len(lat_cold_k), 3) if lat_cold_k else None,
            "value_cold": round(W * H / lat_cold_ms / 1e3, 4) if lat_cold_ms is not None else None,
            "tile_order": (("long tiles first, from the probe launch of the first frame of the view (warm: the probe's order is "
                            "reused; cold: probe launch inside the frame)") if ordered_frames and is_po
                           else ("pixels in the order of the previous frame's counts, longest first (warm: lane s of the launch renders "
                                 "the pixel that ranked s-th; cold: 8 x 8 tiles)") if ordered_frames and (is64 or is2x32)
                           else "longest first, recorded by the previous frame (warm)" if ordered_frames else "natural"),
            "what": "value / ms_per_step = sustained: frames back to back, frame k+1's kernel runs while frame k is gathered and "
                    "copied to the host (two buffers in rotation, copy stream, the host waits on the copy's event).  latency = one "
                    "frame, launch -> host, nothing overlapped.  cold = natural tile order (the first frame of a view), warm = "
                    "tiles launched longest first from the costs the previous frame recorded."}

    # ---- secondary: the same frame with the LA stages in use (GPU-direction stage test), driver-timed like `value`
    secondary = None
    frame_secondary = None
    if wl == "c3_lav2" and args.parity == "cpu" and not args.no_secondary:
        st2 = count_steps(PARITY_CPU_GPUSTAGE)
        one_frame(PARITY_CPU_GPUSTAGE)
        el2, km2 = timed(PARITY_CPU_GPUSTAGE)
        lat2_ms, _ = latency(PARITY_CPU_GPUSTAGE, n_lat)
        if rank == 0:
            b2 = state["last"]
            secondary = {"what": "same frame, FS_PARITY_CPU_GPUSTAGE: CPU arithmetic with the LA stage test in the "
                                 "direction of GPU_LAReference.h:240-254 (the LA stages are used)",
                         "ms_per_step": round(el2 / args.steps * 1e3, 3),
                         "value": round(W * H * args.steps / el2 / 1e6, 4), "unit": "Mpix/s",
                         "latency_ms_warm": round(lat2_ms, 3),
                         "kernel_ms_rank0": round(sum(km2) / len(km2), 3),
                         "la_steps_per_launch": float(st2["la_steps"]),
                         "at_iterations_per_launch": float(st2["at_iterations"]),
                         "pixel_steps_per_launch": float(st2["perturb_steps"]),
                         "frame_checksum": int(host_np[b2][:H, :W].astype(np.uint64).sum())}
            frame_secondary = host_np[b2].copy()
            orc2 = oracle_frame_record(inp["key"], "cpu_gpustage", n_iter)
            secondary["frame_crc32"] = frame_crc32(frame_secondary, H, W)
            secondary["frame_crc32_equals_oracle_frame"] = (None if orc2 is None else
                                                             bool(orc2["crc32"] == secondary["frame_crc32"]))

    # ---- CPU baseline on a bounded sample of rows (rank 0, N = 1 only)
    cpu_baseline = None
    cpu_baseline_patched = None
    parity_rows_ok = None
    if rank == 0 and not args.no_cpu:
        import _oracle
        threads = effective_cpus()
        # rows per host thread chosen per workload so that the sample is about 10 s of CPU work on the box's 16 threads (the
        # GPU part of the default run is ~2 s: a longer CPU leg hides it from a utilisation sampler); at N > 1 the same
        # check runs on rank 0 with one row per thread: the line of an N-GPU run carries its own parity verdict
        per_thread = {"c1_direct": 96, "c3_lav2": 2, "c2_po": 1, "c5_bla": 12, "c4_hdr64": 24, "c4_2x32": 3, "c4_scaled": 2}[wl]
        if distributed:
            per_thread = 1
        nrows = max(1, min(args.cpu_sample_rows if args.cpu_sample_rows > 0 else per_thread * threads, H))
        step = max(1, H // nrows)
        y0 = step // 2
        rows = list(range(y0, H, step))
        _oracle.lib()  # build / load outside the timed window

        def cpu_run(stage_test):
            t1 = time.perf_counter()
            ref = _oracle.workload_rows(inp, y0, H, threads=threads, stage_test=stage_test, row_step=step)
            dt = time.perf_counter() - t1
            return ref, dt

        model, nproc = cpu_model()

        def line(dt, what):
            return {"value": round(len(rows) * W / dt / 1e6, 6), "unit": "Mpix/s", "cores": threads,
                    "cpu_model": model, "nproc": nproc, "kind": "port", "what": what,
                    "compiler": _oracle.compiler_and_flags(),
                    "sample": "%d of %d rows of the same %dx%d frame (rows spread evenly), %.1f s" %
                              (len(rows), H, W, H, dt)}

        ref, cpu_t = cpu_run(0 if args.parity == "cpu" else 1)
        parity_rows_ok = all(np.array_equal(frame_main[y], ref[y]) for y in rows)
        twin = {"c1_direct": "Cpu64 (CalcCpuHDR<uint32_t,double,double>)", "c3_lav2": "Cpu32PerturbedBLAV2HDR", "c4_hdr64": "Cpu64PerturbedBLAV2HDR",
                "c2_po": "Cpu32PerturbedBLAHDR (single-step branch)",
                "c5_bla": "Cpu32PerturbedBLAHDR"}.get(wl, "the CUDA kernel (no CPU twin exists)")
        if not distributed:  # (a reported baseline at N = 1 only; at N > 1 the rows are the line's parity check)
            cpu_baseline = line(cpu_t, "oracle restatement of " + twin +
                                (", unmodified" if args.parity == "cpu" else ", LA stage test patched to the GPU direction"))
        if secondary is not None and not distributed:
            ref2, cpu_t2 = cpu_run(1)
            secondary["cpu_sample_rows_bit_exact"] = all(np.array_equal(frame_secondary[y], ref2[y]) for y in rows)
            cpu_baseline_patched = line(cpu_t2, "the same function with LAReference::isLAStageInvalid in the GPU direction")

    traffic = None
    if rank == 0:
        # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (collected in separate
        # --pmc runs by tools/pmc_passes.sh); null for workloads that have not been profiled.
        key = "view%d_%dx%d_%s" % (args.view, W, H, wl_tag)
        # this round's file only: a workload that has not been profiled with the kernels of this round prints null, never
        # another round's number
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            if key in tj and not distributed and args.parity == inp_default_parity and args.iter_cap == 0:
                traffic = tj[key]["traffic_bytes"]
        except (OSError, ValueError, KeyError):
            pass
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = W * H * args.steps / elapsed / 1e6
        flops_issued = None
        if is2x32:
            # SURVEY.md 8(d)'s ALGORITHMIC count, the same for every numeric type: AT iteration 10, perturbation step 18, LA step
            # 40 flop.  Beside it (frac_issued_ops) the binary32 operations the double-float sequences issue for them (DESIGN.md
            # 4.4): product = 12 flop (1 mul, 4 fma, 3 add), sum = 20 flop; AT iteration = 8 products + 5 sums, perturbation
            # step = 12 products + 10 sums + 6 exact power-of-two scalings (another 6 products), LA step = 22 products + 12 sums.
            flops = at_iters_executed * 10.0 + perturb_steps * FLOP_PER_STEP + la_steps * 40.0
            flops_issued = at_iters_executed * 196.0 + perturb_steps * 416.0 + la_steps * 504.0
        elif is64:
            # HDRFloat<double>: FP64 operations; AT iteration = 5 mul + 5 add (z*z + c on bare mantissas + the norm),
            # perturbation step = 18, LA step = 2 complex mul-adds + 2 norms = 40
            flops = at_iters_executed * 10.0 + perturb_steps * FLOP_PER_STEP + la_steps * 40.0
        elif is_scaled:
            # scaled kernel counters: [0] rescales, [1] full-precision (HDRFloat) steps, [2] binary32 steps; every step is
            # the 18-flop perturbation step of SURVEY 8(d)
            flops = (perturb_steps + la_steps) * FLOP_PER_STEP
        elif is_direct:
            flops = perturb_steps * FLOP_PER_DIRECT_ITERATION  # (the direct kernel counts its iterations in the step counter)
        elif wl == "c5_bla":
            # a BLA jump = dz' = A dz + B dc (two complex products, 6 flop each, + one complex sum, 2), z = Z + dz' (2),
            # |z|^2 (3), |dz'|^2 (3) = 22 flop; the kernel's la_steps counter holds the jumps
            flops = perturb_steps * FLOP_PER_STEP + la_steps * FLOP_PER_BLA_JUMP
        else:
            flops = perturb_steps * FLOP_PER_STEP
        peak = PEAK_FP64_VECTOR_TFLOPS if ((is64 and not is2x32) or is_direct) else PEAK_FP32_VECTOR_TFLOPS
        achieved = flops / (avg_kernel_ms * 1e-3) / 1e12
        roof = {"bound": "valu", "achieved": round(achieved, 4), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 5), "traffic": traffic}
        if at_iters_executed != at_iters:
            roof["at_iterations_executed_per_launch"] = at_iters_executed
            roof["what"] = ("frac prices the work the launch RAN: at_iterations_executed (the cycle search of the AT loop spares pixels "
                            "inside the set most of the at_iterations_per_launch the reference's loop runs for the same frame), "
                            "perturbation and LA steps as counted")
        if flops_issued is not None:
            roof["frac_issued_ops"] = round(flops_issued / (avg_kernel_ms * 1e-3) / 1e12 / peak, 5)
            roof["what"] = ("frac = algorithmic flop (SURVEY 8(d): AT 10 / step 18 / LA step 40) / kernel time / FP32 vector peak; "
                            "frac_issued_ops = the binary32 operations of the double-float sequences that carry them out")
        if wl == "c5_bla":
            # C5 is priced against HBM: achieved = ALGORITHMIC bytes per launch (compulsory traffic: the
            # prepared orbit once, the table once, the iteration buffer written once) / kernel time against 8 TB/s, with the
            # counter bytes beside it (`traffic`; traffic >> algorithmic = re-reads from L2 misses).  The vector-issue view
            # of the same launch is kept under "valu".
            alg_bytes = orbit.count * 16 + sum(bla.sizes()) * 44 + rows_padded * rw * 4  # (orbit + table + iteration buffer, once each)
            gbs = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(gbs, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 6), "traffic": traffic, "algorithmic_bytes": alg_bytes,
                    "traffic_gbs": round(traffic / (avg_kernel_ms * 1e-3) / 1e9, 2) if traffic else None,
                    "valu": {"achieved": round(achieved, 4), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(achieved / peak, 5), "flop_per_bla_jump": FLOP_PER_BLA_JUMP,
                             "bla_jumps_per_launch": la_steps}}
        wl_key = "view%d_%dx%d_%s" % (args.view, W, H, wl_tag)
        orc = oracle_frame_record(wl_key, args.parity, n_iter)
        out = {
            "metric": "Mpix/s (iteration buffer), View #5 3840x2160 HDRx32 LAv2" if wl == "c3_lav2"
            else "Mpix/s (iteration buffer), " + wl,
            "value": round(value, 4), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            # a view's FIRST frame beside the sustained figure (frame_timing.latency_ms_cold: one frame, launch -> host, nothing
            # overlapped, nothing recorded by an earlier frame); for the two workloads whose `value` is cold already, the warm
            # sustained figure and its kernel time
            "value_cold": (round(value, 4) if cold_headline else
                           frame_timing["value_cold"] if frame_timing else None),
            "value_warm": round(W * H * args.steps / elapsed_warm / 1e6, 4) if cold_headline else round(value, 4),
            "kernel_ms_warm": (round(sum(kernel_ms_warm) / len(kernel_ms_warm), 3) if cold_headline else round(avg_kernel_ms, 3)),
            "kernel_parts_ms_warm": ({"k_at_pass64": round(sum(warm_split[0]) / len(warm_split[0]), 3),
                                      "k_lav2_hdr64": round(sum(warm_split[1]) / len(warm_split[1]), 3)}
                                     if cold_headline and wl == "c4_hdr64" and any(warm_split[0]) else None),
            "vs_baseline": None,
            "dtype": "2xf32+i32exp" if is2x32 else ("f64+i32exp" if is64 else ("f64" if is_direct else "f32+i32exp")),
            "data": "built-in view %d of the reference (deterministic: no dataset or randomness on this path)" % args.view,
            "window": "kernel launch -> iteration buffer in (page-locked) host memory, SURVEY.md 8(d); K frames back to back "
                      "(sustained, see frame_timing for the latency of one frame)",
            "config": {"workload": "view%d_%dx%d_%s" % (args.view, W, H, wl_tag),
                       "parity": args.parity, "kernel_variant": args.variant or "tuned", "n_iterations": n_iter,
                       "orbit_entries": orbit.count if orbit is not None else 0,
                       "host_path": ("one device" if not distributed else args.host_path),
                       "frames": ("cold: every timed frame is the first frame of its view (fs_forget_tile_costs before each launch)"
                                  if cold_headline else "warm: frames of one view back to back (value_cold = a view's first frame)"),
                       "la_records": la.count if la else 0, "la_stages": la.stage_count if la else 0,
                       "bla_levels": bla.num_levels if bla else 0, "tiling": "rows/8-row bands interleaved x%d" % world,
                       "exchange": ("none" if not distributed else
                                    "none on the data path: every rank copies its bands to the shared host frame over its own PCIe link"
                                    if direct_path else "gloo gather through host memory (functional path)" if host_staged
                                    else "RCCL gather to rank 0 on the device"),
                       "host_input_build_s": round(t_inputs, 3), "la_build_on_device_ms": la_device_ms},
            "roofline": {**roof,
                         "kernel": (wl_kernel if wl != "c4_hdr64" else
                                    "k_lav2_hdr64<0, false, true> (a view's first frame: ONE kernel, PerformAT iterated inside, tile mapping)"
                                    if cold_headline else "k_at_pass64 + k_lav2_hdr64<0, false, false> (AT pass, then the frame's kernel)"),
                         "kernel_ms": round(avg_kernel_ms, 3),
                         "kernel_parts_ms": ({"k_at_pass64": round(sum(kernel_split[0]) / len(kernel_split[0]), 3),
                                              "k_lav2_hdr64": round(sum(kernel_split[1]) / len(kernel_split[1]), 3)}
                                             if wl == "c4_hdr64" and any(kernel_split[0]) else None),
                         "pixel_steps_per_launch": perturb_steps, "at_iterations_per_launch": at_iters,
                         "la_steps_per_launch": la_steps, "flop_per_pixel_step": FLOP_PER_STEP,
                         "careful_step_fraction_rank0": round(st.get("careful_steps", 0) / max(1.0, perturb_steps), 5)
                         if not distributed else None,
                         "stat6_rank0": st.get("scaled_steps"), "stat7_rank0": st.get("scaled_runs"),
                         "lane_utilisation_rank0": round(perturb_steps / lane_slots, 4)
                         if lane_slots and not distributed else None},
            "cpu_baseline": cpu_baseline, "cpu_baseline_patched": cpu_baseline_patched, "secondary": secondary,
            "frame_checksum": checksum, "cpu_sample_rows_bit_exact": parity_rows_ok,
            # the WHOLE frame against the CPU oracle's frame (tests/golden/frame_crcs.json: rendered by the oracle in the build
            # container, no GPU involved): CRC-32 of the valid region; null when no oracle frame is committed for this
            # workload / parity / cap
            "frame_crc32": crc_main,
            "frame_crc32_equals_oracle_frame": None if orc is None else bool(orc["crc32"] == crc_main),
            "frame_checksum_equals_oracle_frame": None if orc is None else bool(int(orc["sum"]) == int(checksum)),
            "frame_timing": frame_timing,
            "device_resident_ms": round(avg_kernel_ms, 3),
            "device_resident_mpix_s": round(W * H / avg_kernel_ms / 1e3, 4),
        }
        result_line = json.dumps(out)
    if direct_path:
        torch.cuda.synchronize()
        dist.barrier()
        lib.fs_host_unregister(shm_maps.ctypes.data)
        host_np = host_ptr = flags = None
        if rank == 0:
            try:
                os.unlink(shm_files[0])
            except OSError:
                pass
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (result_line + "\n").encode())


if __name__ == "__main__":
    main()
