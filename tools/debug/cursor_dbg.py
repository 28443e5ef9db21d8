import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import numpy as np
    from fractalshark_amd import GPURenderer, T_HDR32, T_HDR64
    from test_gpu_wide_positions import _waypoints
    is64, which, wide, start, n = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    r = GPURenderer(0); lib = r._lib
    T = T_HDR64 if is64 else T_HDR32
    base = (1 << 32) - 40
    big = [0, 100, base + 10, base + 37, base + 40 + 9, base + 40 + 300, (1 << 33) + 5]
    shift = base - 1000
    small = [0, 100] + [i - shift for i in big[2:]]
    idx = big if which == "big" else small
    cdt = np.dtype([("re", "<f8"), ("im", "<f8"), ("e", "<i4"), ("pad", "<i4")]) if is64 else np.dtype([("re", "<f4"), ("im", "<f4"), ("e", "<i4")])
    low = np.zeros(2, np.dtype([("m", "<f8"), ("e", "<i4"), ("pad", "<i4")]) if is64 else np.dtype([("m", "<f4"), ("e", "<i4")]))
    low["m"] = [-1.25, 1.5]; low["e"] = [-1, -3]
    assert r.set_compressed_orbit_mode(True) == 0
    assert r.InitializeMemory(64, 36, 1, None, 0, 0, 0, False, iter_bytes=8) == 0
    wp = _waypoints(bool(is64), idx, 7)
    print("upload", lib.fs_upload_orbit_compressed(r._h, 0, T, 8, wp.ctypes.data, len(wp), int(idx[-1]) + 50, 0, low[0:1].ctypes.data, low[1:2].ctypes.data), flush=True)
    out = np.zeros(n, cdt)
    print("probe", lib.fs_seq_cursor_probe(r._h, wide, start, n, out.ctypes.data), flush=True)
    print(out[:4], flush=True)
else:
    base = (1 << 32) - 40; shift = base - 1000
    for is64 in (0, 1):
        for which, wide, start, n in (("big", 1, base + 10, 8), ("small", 0, 1010, 8), ("small", 1, 1010, 8), ("big", 1, base + 3, 70), ("small", 0, 1003, 70)):
            try:
                p = subprocess.run([sys.executable, __file__, str(is64), which, str(wide), str(start), str(n)], timeout=25, capture_output=True, text=True)
                print(is64, which, wide, start, n, "->", p.returncode, p.stdout.replace("\n", " | ")[-300:], p.stderr[-200:].replace("\n"," | "), flush=True)
            except subprocess.TimeoutExpired:
                print(is64, which, wide, start, n, "-> TIMEOUT", flush=True)
