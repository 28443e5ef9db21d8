"""The integer stream of Imagina ".im" files (the two mpf values of a location: an exponent word, then the mantissa as an MPIR
raw integer) against the byte vectors of the reference's OWN unit tests for it (FractalSharkTest/TestMpirSerialization.cpp:
MpirSer_WireFormat_*, MpirSer_GoldenBinary_CrossFormat -- tests/golden/mpir_wire_vectors.json, extracted as data by
tests/golden/make_mpir_wire_vectors.py).  This is the one part of the ".im" path a reference-held vector pins: the writer
must produce exactly these bytes and the reader must give these values back (host library, no GPU)."""
import ctypes as C
import json
import os
import random

import pytest

from fractalshark_amd import _capi

VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpir_wire_vectors.json")))


@pytest.fixture(scope="module")
def lib(native_libs):
    return _capi.inputs_lib()


def write(lib, value, base=10):
    buf = C.create_string_buffer(1 << 16)
    n = lib.fsh_mpz_raw_write(str(value).encode(), base, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def read(lib, data):
    out = C.create_string_buffer(1 << 16)
    used = lib.fsh_mpz_raw_read(bytes(data), len(data), out, len(out))
    assert used > 0
    return int(out.value.decode()), used


@pytest.mark.parametrize("case", VEC["wire_format"], ids=[c["name"] for c in VEC["wire_format"]])
def test_wire_format_vectors(lib, case):
    want = bytes.fromhex(case["bytes"])
    assert write(lib, case["value"], case["base"]) == want, case["source"]
    value, used = read(lib, want)
    assert used == len(want) and value == int(case["value"], case["base"])


def test_golden_stream_both_directions(lib):
    g = VEC["golden_stream"]
    blob = bytes.fromhex(g["bytes"])
    # read: the ten integers come back in order and the stream is consumed exactly
    at, got = 0, []
    while at < len(blob):
        v, used = read(lib, blob[at:])
        got.append(v)
        at += used
    assert got == [int(d) for d in g["decimals"]] and at == len(blob)
    # write: the concatenated streams are the reference's bytes
    assert b"".join(write(lib, d) for d in g["decimals"]) == blob


def test_round_trip_of_random_integers(lib):
    """MpirSer_MpzRoundtrip_100Random's shape: multi-limb values of varied width, signs and zeros, one after the other in one
    stream (own generator: the reference's seeds an mt19937_64, whose draws a Python port would have to restate)."""
    rng = random.Random(0xBADC0FFEE123)
    vals = []
    for _ in range(100):
        v = 0
        for _ in range(1 + rng.randrange(32)):
            v = (v << 64) + rng.getrandbits(64)
        if rng.getrandbits(1):
            v = -v
        if rng.getrandbits(4) == 0:
            v = 0
        vals.append(v)
    blob = b"".join(write(lib, v) for v in vals)
    at, got = 0, []
    while at < len(blob):
        v, used = read(lib, blob[at:])
        # header = sign x byte count, big-endian; magnitude most significant byte first
        n = int.from_bytes(blob[at:at + 4], "big", signed=True)
        assert used == 4 + abs(n) and (n > 0) == (v > 0) and (n == 0) == (v == 0)
        assert int.from_bytes(blob[at + 4:at + used], "big") == abs(v)
        got.append(v)
        at += used
    assert got == vals


def test_truncated_and_oversized_streams_are_refused(lib):
    out = C.create_string_buffer(64)
    assert lib.fsh_mpz_raw_read(b"\x00\x00", 2, out, len(out)) == 0                       # no header
    assert lib.fsh_mpz_raw_read(b"\x00\x00\x00\x05\x01", 5, out, len(out)) == 0           # magnitude cut short
    assert lib.fsh_mpz_raw_read(b"\x7f\xff\xff\xff\x01", 5, out, len(out)) == 0           # absurd byte count
    assert lib.fsh_mpz_raw_write(b"12x", 10, out, len(out)) == 0                          # not a number
