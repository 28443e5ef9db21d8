"""Imagina ".im" files that carry a reference orbit: the writer
(RefOrbitCalc::SaveOrbitResults(results, filename) -> CompressMax + SaveOrbitBin) and the reader (LoadOrbitBin +
DecompressMax) restated in fractalshark_amd/host/refinputs.cpp.

Parity status: UNPINNED against reference-written files (the reference tree's .im files are git-LFS pointer stubs).
Pinned here: the byte layout -- restated independently below, and the struct sizes / offsets it assumes are checked
against the reference's real headers when the tree is present -- and the property the format exists for: the orbit a
reader rebuilds from a few hundred bytes of waypoints is the orbit that was saved, to the compression tolerance.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from fractalshark_amd import inputs

IM_MAGIC, SHARKS_MAGIC = 0x000A0D56504D49FF, 0x536861726b733a29
REF = "/root/reference"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def values(e):
    x = e["mx"].astype(np.float64) * np.exp2(e["ex"].astype(np.float64))
    y = e["my"].astype(np.float64) * np.exp2(e["ey"].astype(np.float64))
    return x, y


def parse(path):
    b = open(path, "rb").read()
    magic, reserved, loc, ref = struct.unpack("<4Q", b[:32])
    mant, e, limit = struct.unpack("<dqQ", b[loc:loc + 24])
    at = ref
    ext = b[at]
    at += 1
    trivial = [struct.unpack("<dq", b[at + 16 * k:at + 16 * k + 16]) for k in range(3)]
    at += 48
    la = b[at:at + 192]
    at += 192
    (n,) = struct.unpack("<Q", b[at:at + 8])
    at += 8
    wps = []
    for _ in range(n):
        xm, xe, ym, ye, field = struct.unpack("<dqdqQ", b[at:at + 40])
        wps.append((xm, xe, ym, ye, field & (2 ** 63 - 1), field >> 63))
        at += 40
    (r,) = struct.unpack("<Q", b[at:at + 8])
    at += 8
    rebases = list(struct.unpack("<%dQ" % r, b[at:at + 8 * r]))
    at += 8 * r
    return dict(magic=magic, reserved=reserved, loc=loc, ref=ref, halfH=(mant, e), limit=limit, ext=ext, trivial=trivial,
                refc=struct.unpack("<2d", la[0:16]), ref_it=struct.unpack("<Q", la[16:24])[0],
                max_it=struct.unpack("<Q", la[24:32])[0], flags=tuple(la[32:36]), at_block=la[40:184],
                la_stage_count=struct.unpack("<Q", la[184:192])[0], waypoints=wps, rebases=rebases, end=at, size=len(b))


@pytest.mark.parametrize("is64", [True, False])
@pytest.mark.parametrize("exp_bytes", [4, 8])
def test_file_layout(tmp_path, is64, exp_bytes):
    v = inputs.View.builtin(5, 64, 36)
    o = inputs.Orbit(v, is64=is64)
    p = tmp_path / "o.im"
    o.save_im(p, exp_bytes=exp_bytes)
    f = parse(p)
    assert f["magic"] == (IM_MAGIC if is64 else SHARKS_MAGIC)  # SubType double <-> Imagina's magic, float <-> "Sharks:)"
    assert (f["reserved"], f["loc"]) == (0, 32) and f["ref"] > 32 + 24 and f["end"] == f["size"]
    assert f["limit"] == v.num_iterations  # GetMaxIterations() - 1 with MaxIterations = NumIterations + 1
    assert f["ext"] == 1
    # ReferenceTrivialContent: {2, -precision}, {}, ValidRadius = MaxRadius = halfH
    assert f["trivial"][0][0] == 2.0 and f["trivial"][0][1] < 0 and -f["trivial"][0][1] >= v.precision_bits
    assert f["trivial"][1] == (0.0, 0)
    assert f["trivial"][2] == f["halfH"]
    mr = o.max_radius()[0]
    assert f["halfH"] == (float(mr["m"]), int(mr["e"]))
    # LAReferenceTrivialContent
    assert f["ref_it"] == o.count - 1 and f["max_it"] == v.num_iterations - 1
    assert f["flags"] == (0, 0, 1 if o.period else 0, 0) and f["at_block"] == bytes(144) and f["la_stage_count"] == 0
    # waypoints: strictly increasing orbit indices inside the orbit, finite mantissas below 2 (orbit values, or differences
    # of orbit values that the reference stores without HdrReduce)
    idx = [w[4] for w in f["waypoints"]]
    assert idx and idx == sorted(set(idx)) and 1 <= idx[0] and idx[-1] < o.count
    for xm, xe, ym, ye, _, _ in f["waypoints"]:
        assert abs(xm) < 2.0 and abs(ym) < 2.0
    assert all(0 < r < o.count for r in f["rebases"])
    # the point of the format: a long orbit in very few bytes (HDRFloat<double>: the orbit re-derives itself)
    if is64:
        assert f["size"] < 4096 and o.count > 10000


@pytest.mark.parametrize("view_n,is64,tol", [(5, True, 1e-10), (11, True, 1e-10), (19, True, 1e-10), (3, True, 1e-10),
                                              (5, False, 2e-3), (3, False, 2e-3)])
def test_round_trip_rebuilds_the_orbit(tmp_path, view_n, is64, tol):
    v = inputs.View.builtin(view_n, 64, 36)
    o = inputs.Orbit(v, is64=is64)
    p = tmp_path / "o.im"
    o.save_im(p)
    w = inputs.View.load_im(p, 64, 36)
    assert w.im_has_orbit
    q = inputs.Orbit.load_im(p, w)
    assert (q.count, q.period, q.is64) == (o.count, o.period, is64)
    assert q.im_iteration_limit == v.num_iterations
    ax, ay = values(o.entries())
    bx, by = values(q.entries())
    cheb = np.maximum(np.abs(ax), np.abs(ay))
    err = np.maximum(np.abs(ax - bx), np.abs(ay - by))
    assert (err[1:] <= tol * cheb[1:]).all(), float((err[1:] / cheb[1:]).max())
    assert ax[0] == bx[0] == 0.0 and ay[0] == by[0] == 0.0
    # same low-precision reference point and radius as the orbit that was saved
    assert np.array_equal(o.orbit_low(), q.orbit_low())
    assert np.array_equal(o.max_radius(), q.max_radius())


def test_files_without_extended_range_or_without_an_orbit_are_refused(tmp_path):
    v = inputs.View.builtin(5, 32, 32)
    o = inputs.Orbit(v, is64=True)
    p = tmp_path / "o.im"
    o.save_im(p)
    raw = bytearray(open(p, "rb").read())
    ref = struct.unpack("<Q", raw[24:32])[0]
    plain = bytearray(raw)
    plain[ref] = 0  # ReferenceHeader.ExtendedRange = false: a plain double orbit, which this reader does not take
    q = tmp_path / "plain.im"
    q.write_bytes(bytes(plain))
    with pytest.raises(ValueError):
        inputs.Orbit.load_im(q, v)
    cut = tmp_path / "cut.im"
    cut.write_bytes(bytes(raw[:-5]))
    with pytest.raises(ValueError):
        inputs.Orbit.load_im(cut, v)
    loc_only = tmp_path / "loc.im"
    v.save_im(loc_only)
    with pytest.raises(ValueError):
        inputs.Orbit.load_im(loc_only, v)


@pytest.mark.parametrize("is64", [True, False])
def test_corrupt_or_crafted_orbit_sections_are_refused_not_fatal(tmp_path, is64):
    """A file is untrusted input: an orbit length that does not fit (it sizes the reader's vectors), a waypoint at index 0
    with the rebase bit (the reader would look at the entry before it), waypoints out of order or past the orbit, and
    exponents beyond int32 all come back as a refusal -- the process survives and a good file still loads afterwards."""
    v = inputs.View.builtin(5, 32, 32)
    o = inputs.Orbit(v, is64=is64)
    p = tmp_path / "o.im"
    o.save_im(p)
    raw = bytearray(open(p, "rb").read())
    info = parse(p)
    ref = info["ref"]
    assert len(info["waypoints"]) >= 2
    la_at = ref + 1 + 48            # ReferenceHeader: ExtendedRange byte, three HRReal, then the LA block
    refit_at = la_at + 16
    wp0 = la_at + 192 + 8           # first waypoint: x (16 B), y (16 B), index | rebase << 63
    assert struct.unpack("<Q", raw[refit_at:refit_at + 8])[0] == info["ref_it"] == o.count - 1

    def refused(name, patch):
        b = bytearray(raw)
        patch(b)
        q = tmp_path / (name + ".im")
        q.write_bytes(bytes(b))
        with pytest.raises(ValueError):
            inputs.Orbit.load_im(q, v)

    refused("refit_huge", lambda b: b.__setitem__(slice(refit_at, refit_at + 8), struct.pack("<Q", 2 ** 40)))
    refused("refit_all_ones", lambda b: b.__setitem__(slice(refit_at, refit_at + 8), struct.pack("<Q", 2 ** 64 - 1)))
    refused("refit_past_limit", lambda b: b.__setitem__(slice(refit_at, refit_at + 8), struct.pack("<Q", info["limit"] + 1)))
    refused("wp_index0_rebase", lambda b: b.__setitem__(slice(wp0 + 32, wp0 + 40), struct.pack("<Q", 1 << 63)))
    refused("wp_not_increasing", lambda b: b.__setitem__(slice(wp0 + 40 + 32, wp0 + 80), b[wp0 + 32:wp0 + 40]))
    refused("wp_past_orbit", lambda b: b.__setitem__(slice(wp0 + 32, wp0 + 40), struct.pack("<Q", info["ref_it"] + 5)))
    refused("exp_beyond_int32", lambda b: b.__setitem__(slice(wp0 + 8, wp0 + 16), struct.pack("<q", 2 ** 40)))
    # the rebase count (8 bytes behind the last waypoint) sizes a vector too: 2^32 of them (32 GiB, zero-filled) used to be
    # allocated before the short read was noticed; more rebases than orbit entries is refused before any allocation
    reb_at = wp0 + 40 * len(info["waypoints"])
    assert struct.unpack("<Q", raw[reb_at:reb_at + 8])[0] == len(info["rebases"])
    refused("rebases_2_32", lambda b: b.__setitem__(slice(reb_at, reb_at + 8), struct.pack("<Q", 2 ** 32)))
    refused("rebases_past_orbit", lambda b: b.__setitem__(slice(reb_at, reb_at + 8), struct.pack("<Q", info["ref_it"] + 2)))
    refused("rebases_short_read", lambda b: b.__setitem__(slice(reb_at, reb_at + 8), struct.pack("<Q", len(info["rebases"]) + 3)))
    q = inputs.Orbit.load_im(p, v)
    assert q.count == o.count


LAYOUT_PROBE = r"""
#include <cstddef>
#include <cstdio>
#include "ImaginaOrbit.h"
#include "GPU_ReferenceIter.h"
int main() {
    using namespace Imagina;
    HRReal p{-(int64_t)1234, 2};
    CompressionIndexField c(5, 1);
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %g %lld %llx\n", sizeof(IMFileHeader), sizeof(HRReal),
           sizeof(ReferenceHeader), sizeof(ReferenceTrivialContent), sizeof(ImaginaATInfo), sizeof(LAReferenceTrivialContent),
           offsetof(LAReferenceTrivialContent, RefIt), offsetof(LAReferenceTrivialContent, MaxIt),
           offsetof(LAReferenceTrivialContent, IsPeriodic), offsetof(LAReferenceTrivialContent, AT),
           offsetof(LAReferenceTrivialContent, LAStageCount), sizeof(CompressionIndexField), (double)p.getMantissa(),
           (long long)p.getExp(), (unsigned long long)c.u.Raw);
    return 0;
}
"""
FORMAT_STUB = """#pragma once
#include <string>
namespace std {
template <class... A> std::string format(const char *, A &&...) { return {}; }
template <class... A> std::string format(const std::string &, A &&...) { return {}; }
}
"""


@pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "FractalSharkLib")) and os.path.exists(CLANG)),
                    reason="needs the reference tree and ROCm clang++")
def test_layout_constants_against_the_reference_headers(tmp_path):
    """The sizes and offsets the writer / reader hard-code, from the reference's own ImaginaOrbit.h and
    GPU_ReferenceIter.h (compiled as they lie; the only thing written for the compile is a <format> header for
    libstdc++ 11, as in tests/test_shim_real_headers.py)."""
    stub = tmp_path / "stdstub"
    stub.mkdir()
    (stub / "format").write_text(FORMAT_STUB)
    src = tmp_path / "probe.cpp"
    src.write_text(LAYOUT_PROBE)
    exe = tmp_path / "probe"
    cmd = [CLANG, "-std=c++23", "-Wno-everything", "-I" + str(stub), "-I/opt/conda/include", "-I" + REF + "/HpSharkFloatLib",
           "-I" + REF + "/FractalSharkPlatform/Common", "-I" + REF + "/FractalSharkLib", "-I" + REF, str(src), "-o", str(exe),
           "-L/opt/conda/lib", "-lgmp", "-Wl,-rpath,/opt/conda/lib"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-3000:]
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert [int(x) for x in out[:12]] == [32, 16, 1, 48, 144, 192, 16, 24, 34, 40, 184, 8]
    assert (float(out[12]), int(out[13])) == (2.0, -1234)  # HRReal{exp, mantissa}: the AbsolutePrecision field
    assert out[14] == "8000000000000005"                   # 63-bit index, rebase flag in the top bit


# ---- the non-ExtendedRange forms: PerturbationResults<IterType, float | double, ...> (round 4)
def parse_plain(path):
    b = open(path, "rb").read()
    magic, reserved, loc, ref = struct.unpack("<4Q", b[:32])
    mant, e, limit = struct.unpack("<dqQ", b[loc:loc + 24])
    at = ref
    ext = b[at]
    at += 1 + 48
    la = b[at:at + 192]
    at += 192
    (n,) = struct.unpack("<Q", b[at:at + 8])
    at += 8
    wps = []
    for _ in range(n):
        x, y, field = struct.unpack("<ddQ", b[at:at + 24])  # double x, double y, CompressionIndexField
        wps.append((x, y, field & (2 ** 63 - 1), field >> 63))
        at += 24
    (r,) = struct.unpack("<Q", b[at:at + 8])
    at += 8 + 8 * r
    return dict(magic=magic, loc=loc, ref=ref, halfH=(mant, e), limit=limit, ext=ext, ref_it=struct.unpack("<Q", la[16:24])[0],
                max_it=struct.unpack("<Q", la[24:32])[0], flags=tuple(la[32:36]), waypoints=wps, end=at, size=len(b))


def _shallow(width="1e-5", n_iter=20000):
    from test_plain_oracle import shallow_view
    return shallow_view(width, n_iter=n_iter)


@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_plain_file_layout(tmp_path, kind):
    v = _shallow()
    pin = inputs.PlainInputs(v, kind)
    p = tmp_path / "p.im"
    pin.save_im(p)
    f = parse_plain(p)
    assert f["magic"] == (SHARKS_MAGIC if kind == "f32" else IM_MAGIC)  # the magic follows the SubType
    assert f["ext"] == 0                                                 # ReferenceHeader::ExtendedRange = results.IsHDR
    assert f["loc"] == 32 and f["end"] == f["size"]                      # 24-byte waypoints account for every byte
    assert f["limit"] == v.num_iterations and f["ref_it"] == pin.count - 1 and f["max_it"] == v.num_iterations - 1
    assert f["flags"][2] == (1 if pin.period else 0)
    # halfH = Imagina::HRReal{T radius}: normalised mantissa in [1, 2)
    assert 1.0 <= abs(f["halfH"][0]) < 2.0
    idx = [w[2] for w in f["waypoints"]]
    assert idx == sorted(idx) and idx[0] >= 1 and idx[-1] <= pin.count - 1
    if kind == "f32":  # waypoints are floats widened to double: no bits below binary32
        for x, y, _, _ in f["waypoints"]:
            assert np.float64(np.float32(x)) == x and np.float64(np.float32(y)) == y


@pytest.mark.parametrize("kind,width,tol", [("f64", "1e-5", 1e-9), ("f64", "1e-10", 1e-9), ("f32", "1e-4", 2e-3), ("f32", "1e-5", 2e-3)])
def test_plain_round_trip_rebuilds_the_orbit_and_its_table(tmp_path, kind, width, tol):
    v = _shallow(width)
    pin = inputs.PlainInputs(v, kind)
    p = tmp_path / "p.im"
    pin.save_im(p)
    w = inputs.View.load_im(p, v.width, v.height)
    assert w.im_has_orbit
    q = inputs.PlainInputs.load_im(p, w)
    assert (q.kind, q.count, q.period) == (kind, pin.count, pin.period)
    assert q.im_iteration_limit == v.num_iterations
    a, b = pin.orbit(), q.orbit()
    ax, ay, bx, by = (z.astype(np.float64) for z in (a["x"], a["y"], b["x"], b["y"]))
    cheb = np.maximum(np.abs(ax), np.abs(ay))
    err = np.maximum(np.abs(ax - bx), np.abs(ay - by))
    assert (err[1:] <= tol * cheb[1:]).all(), float((err[1:] / cheb[1:]).max())
    assert ax[0] == bx[0] == 0.0 and ay[0] == by[0] == 0.0
    # a table comes with the loaded orbit (the same builder ran on it), of the same shape as the saved orbit's
    assert q.is_valid == pin.is_valid and q.stage_count == pin.stage_count
    assert abs(q.la_count - pin.la_count) <= max(2, pin.la_count // 50)


def test_plain_and_extended_range_readers_refuse_each_other(tmp_path):
    v = _shallow()
    pin = inputs.PlainInputs(v, "f64")
    p = tmp_path / "plain.im"
    pin.save_im(p)
    with pytest.raises(ValueError):
        inputs.Orbit.load_im(p, v)  # ExtendedRange = false: not an HDRFloat orbit
    with pytest.raises(ValueError):
        inputs.PlainInputs.load_im(p, v, kind="f32")  # Imagina's magic: a double orbit
    o = inputs.Orbit(v, is64=True)
    h = tmp_path / "hdr.im"
    o.save_im(h)
    with pytest.raises(ValueError):
        inputs.PlainInputs.load_im(h, v)  # ExtendedRange = true
    raw = bytearray(open(p, "rb").read())
    cut = tmp_path / "cut.im"
    cut.write_bytes(bytes(raw[:-7]))
    with pytest.raises(ValueError):
        inputs.PlainInputs.load_im(cut, v)
    # a crafted waypoint index past the announced orbit
    ref = struct.unpack("<Q", raw[24:32])[0]
    first = ref + 1 + 48 + 192 + 8
    bad = bytearray(raw)
    bad[first + 16:first + 24] = struct.pack("<Q", 2 ** 40)
    b = tmp_path / "bad.im"
    b.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        inputs.PlainInputs.load_im(b, v)
