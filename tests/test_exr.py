"""EXR egress (take_amd/exr.py) against files written by the reference's own imwrite (tests/golden/egress/*.exr were
made by oracle/_ref/ref_harness `imwrite` = src/image.cpp:155-176 -> tinyexr SaveEXR, half, from the .f64 images
beside them; generator: oracle/gen_golden.py egress).  Pinned: channel set and order, pixel type, compression choice,
windows, and every half-precision pixel bit for bit — including rounding ties, overflow to inf and denormals."""
import os
import struct

import numpy as np
import pytest

from helpers import GOLD
from take_amd.exr import float_to_half, read_exr, write_exr

CASES = ["zip_40x37", "none_12x9"]


def load(name):
    a = np.fromfile(os.path.join(GOLD, "egress", name + ".f64"), "<f8")
    w, h = int(a[0]), int(a[1])
    return a[2:].reshape(h, w, 3)


@pytest.mark.parametrize("name", CASES)
def test_reference_file_holds_the_halves_our_conversion_predicts(name):
    img = load(name)
    ch, hdr = read_exr(os.path.join(GOLD, "egress", name + ".exr"))
    assert list(ch) == ["B", "G", "R"] and all(v.dtype == np.float16 for v in ch.values())
    want = float_to_half(img.astype(np.float32))
    for c, k in (("R", 0), ("G", 1), ("B", 2)):
        assert np.array_equal(ch[c].view(np.uint16), want[..., k]), c
    assert np.isinf(ch["R"][0, 1]) and ch["B"][0, 0] == np.float16(65504.0)
    # the reference's writer rounds ties up, not to even: 1 + 2^-11 -> 1 + 2^-10, 2049 -> 2050
    assert ch["B"][0, 1] == np.float16(1.0009765625) and ch["B"][1, 0] == np.float16(2050.0)
    rne = img.astype(np.float32).astype(np.float16)
    assert rne[0, 1, 2] == np.float16(1.0) and rne[1, 0, 2] == np.float16(2048.0)  # what round-to-nearest-even would give


@pytest.mark.parametrize("name", CASES)
def test_our_writer_makes_the_same_image_and_header(name, tmp_path):
    img = load(name)
    out = str(tmp_path / "image.exr")
    write_exr(out, img)
    ours, h1 = read_exr(out)
    ref, h0 = read_exr(os.path.join(GOLD, "egress", name + ".exr"))
    for c in ("B", "G", "R"):
        assert np.array_equal(ours[c].view(np.uint16), ref[c].view(np.uint16)), c
    for key in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder"):
        assert h1[key] == h0[key], key
    assert struct.unpack("<f", h1["pixelAspectRatio"][1])[0] == struct.unpack("<f", h0["pixelAspectRatio"][1])[0]


def test_roundtrip_of_a_large_random_image(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.normal(size=(131, 517, 3)) * 10
    out = str(tmp_path / "x.exr")
    write_exr(out, img)
    ch, _ = read_exr(out)
    assert np.array_equal(ch["G"].view(np.uint16), float_to_half(img.astype(np.float32))[..., 1])
    # apart from ties the conversion agrees with IEEE round-to-nearest
    with np.errstate(over="ignore"):
        assert np.mean(ch["G"].view(np.uint16) == img.astype(np.float32).astype(np.float16)[..., 1].view(np.uint16)) > 0.999
