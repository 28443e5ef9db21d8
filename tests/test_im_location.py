"""Imagina ".im" location files (the data format on the caller's side of the path).

Parity status: UNPINNED against reference-written files -- the only .im files under the reference tree are git-LFS pointer
stubs, not data.  What is pinned here is the byte layout, restated independently in Python from the reference's writer
(RefOrbitCalc.cpp:3117-3166, ImaginaOrbit.h:10-24, MpirSerialization.cpp:96-187), and the round trip.
"""
import struct
import sys
from fractions import Fraction

import numpy as np
import pytest

from fractalshark_amd import inputs

IM_MAGIC = 0x000A0D56504D49FF
if hasattr(sys, "set_int_max_str_digits"):
    sys.set_int_max_str_digits(0)  # view 14's coordinates have thousands of digits


def parse_mpf(buf, at, exp_bytes):
    """(value as a Fraction, next offset): long exponent in 64-bit limbs, int32 BE signed byte count, BE magnitude."""
    exp = int.from_bytes(buf[at:at + exp_bytes], "little", signed=True)
    at += exp_bytes
    n = struct.unpack(">i", buf[at:at + 4])[0]
    at += 4
    mag = int.from_bytes(buf[at:at + abs(n)], "big")
    at += abs(n)
    if n == 0:
        return Fraction(0), at
    limbs = (mag.bit_length() + 63) // 64
    v = Fraction(mag) * Fraction(2) ** (64 * (exp - limbs))
    return (-v if n < 0 else v), at


def parse_im(path, exp_bytes):
    buf = open(path, "rb").read()
    magic, reserved, loc, ref = struct.unpack("<4Q", buf[:32])
    mant, e, limit = struct.unpack("<dqQ", buf[loc:loc + 24])
    x, at = parse_mpf(buf, loc + 24, exp_bytes)
    y, at = parse_mpf(buf, at, exp_bytes)
    return dict(magic=magic, reserved=reserved, loc=loc, ref=ref, mant=mant, exp=e, limit=limit, x=x, y=y, end=at,
                size=len(buf))


def frac(s):
    return Fraction(s)


def bbox_fracs(v):
    # "%.Fe" prints every digit the value has at its precision
    return [Fraction(t) for t in v.bbox()]


@pytest.mark.parametrize("exp_bytes", [4, 8])
@pytest.mark.parametrize("n", [0, 5, 10, 14, 22])
def test_written_file_has_the_reference_layout(tmp_path, n, exp_bytes):
    v = inputs.View.builtin(n, 64, 48)
    p = tmp_path / "v.im"
    v.save_im(p, exp_bytes=exp_bytes)
    f = parse_im(p, exp_bytes)
    assert (f["magic"], f["reserved"], f["loc"], f["ref"]) == (IM_MAGIC, 0, 32, 0)
    assert f["end"] == f["size"]
    assert f["limit"] == v.num_iterations
    mn_x, mn_y, mx_x, mx_y = bbox_fracs(v)
    radius = (mx_y - mn_y) / 2
    if radius < Fraction(1, 2 ** 1022):
        # `double{maxY - minY}` underflows: the reference's writer stores a zero halfH for such views ("only relevant
        # with double/float precision and shallow depths", RefOrbitCalc.cpp:3120-3121), and so does this one
        assert f["mant"] == 0.0 and f["exp"] == -4096
        with pytest.raises(ValueError):
            inputs.View.load_im(p, 64, 48)
        return
    # halfH = HDRFloat{double(maxY - minY) / 2}: mantissa in [1, 2), value = the double nearest below the radius
    assert 1.0 <= f["mant"] < 2.0
    half_h = Fraction(f["mant"]) * Fraction(2) ** f["exp"]
    assert half_h <= radius and radius - half_h <= radius * Fraction(1, 2 ** 50)
    # the centre, to the view's working precision
    tol = radius * Fraction(1, 2 ** 100)
    assert abs(f["x"] - (mn_x + mx_x) / 2) <= tol
    assert abs(f["y"] - (mn_y + mx_y) / 2) <= tol


@pytest.mark.parametrize("exp_bytes", [4, 8])
@pytest.mark.parametrize("n", [0, 5, 10, 22])
def test_round_trip_gives_the_same_location(tmp_path, n, exp_bytes):
    v = inputs.View.builtin(n, 64, 64)  # a square window: the reloaded box is pt -+ halfH in both directions
    p = tmp_path / "v.im"
    v.save_im(p, exp_bytes=exp_bytes)
    w = inputs.View.load_im(p, 64, 64)
    assert w.num_iterations == v.num_iterations
    assert (w.im_has_orbit, w.im_exp_bytes) == (False, exp_bytes)
    a, b = bbox_fracs(v), bbox_fracs(w)
    radius = (a[3] - a[1]) / 2
    # halfH is a double: the box comes back to 2^-52 of its size, the centre to the working precision
    for k in range(4):
        assert abs(a[k] - b[k]) <= radius * Fraction(1, 2 ** 50)
    assert abs((a[0] + a[2]) - (b[0] + b[2])) <= radius * Fraction(1, 2 ** 100)
    assert abs((a[1] + a[3]) - (b[1] + b[3])) <= radius * Fraction(1, 2 ** 100)
    # precision = -min(0, halfH.exp) + 120
    f = parse_im(p, exp_bytes)
    assert w.precision_bits >= -min(0, f["exp"]) + 120 - 2
    # saving the reloaded view again: the same halfH and iteration limit, the centre within the reloaded precision
    q = tmp_path / "w.im"
    w.save_im(q, exp_bytes=exp_bytes)
    g = parse_im(q, exp_bytes)
    assert g["limit"] == f["limit"]
    assert abs(Fraction(g["mant"]) * Fraction(2) ** g["exp"] - Fraction(f["mant"]) * Fraction(2) ** f["exp"]) <= \
        radius * Fraction(1, 2 ** 50)
    assert abs(g["x"] - f["x"]) <= radius * Fraction(1, 2 ** 100)


def test_reloaded_view_gives_the_same_kernel_inputs(tmp_path):
    """What the kernels take from a view -- the reference orbit and the per-pixel delta grid -- built from a reloaded
    location: same orbit length and period, entries equal to float precision, dx / dy equal to a double's precision."""
    v = inputs.View.builtin(5, 32, 32, antialiasing=1)
    p = tmp_path / "v5.im"
    v.save_im(p)
    w = inputs.View.load_im(p, 32, 32)
    ov, ow = inputs.Orbit(v), inputs.Orbit(w)
    assert (ov.count, ov.period) == (ow.count, ow.period)
    cv, cw = v.coords_perturb_hdr32(ov), w.coords_perturb_hdr32(ow)
    for k in range(2):  # dx, dy
        a = float(cv[k]["m"]) * 2.0 ** int(cv[k]["e"])
        b = float(cw[k]["m"]) * 2.0 ** int(cw[k]["e"])
        assert abs(a - b) <= abs(a) * 2.0 ** -22


def test_file_with_a_reference_orbit_is_loaded_as_a_location(tmp_path):
    v = inputs.View.builtin(10, 48, 48)
    p = tmp_path / "v.im"
    v.save_im(p, exp_bytes=4)
    raw = bytearray(open(p, "rb").read())
    ref_at = len(raw)
    raw[24:32] = struct.pack("<Q", ref_at)  # ReferenceOffset
    raw += b"\x01" + b"\x00" * 63  # ReferenceHeader{ExtendedRange} and whatever follows: not read
    raw[0:8] = struct.pack("<Q", 0x536861726b733a29)  # "Sharks:)"
    q = tmp_path / "with_orbit.im"
    q.write_bytes(bytes(raw))
    w = inputs.View.load_im(q, 48, 48)
    assert w.im_has_orbit and w.im_exp_bytes == 4
    assert w.num_iterations == v.num_iterations


@pytest.mark.parametrize("damage", ["magic", "truncated", "empty", "missing"])
def test_bad_files_are_refused(tmp_path, damage):
    v = inputs.View.builtin(0, 16, 16)
    p = tmp_path / "v.im"
    v.save_im(p)
    raw = bytearray(open(p, "rb").read())
    if damage == "magic":
        raw[0] ^= 0x55
    elif damage == "truncated":
        raw = raw[:-3]
    elif damage == "empty":
        raw = bytearray()
    q = tmp_path / "bad.im"
    if damage != "missing":
        q.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        inputs.View.load_im(q, 16, 16)
