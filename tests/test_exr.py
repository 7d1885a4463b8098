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


# ------------------------------------------------------------------ device-side egress
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("precision", [0, 1])
def test_device_packs_the_scanlines_the_reference_writer_stores(name, precision, tmp_path):
    """take_hip_pack_exr_scanlines on the golden float images == the halves inside the EXR files the reference's own
    imwrite made of them, bit for bit (ties, overflow to inf, denormals included); framed by write_exr_scanlines the
    result is the file our host writer produces."""
    import ctypes as C

    import torch

    from take_amd import capi
    from take_amd.exr import write_exr_scanlines

    img = load(name)
    h, w = img.shape[:2]
    dev = torch.tensor(img, dtype=torch.float64 if precision else torch.float32, device="cuda")
    out = torch.zeros((h, 3, w), dtype=torch.int16, device="cuda")
    rc = capi.lib().take_hip_pack_exr_scanlines(C.c_void_p(dev.data_ptr()), precision, w, h, C.c_void_p(out.data_ptr()), None)
    assert rc == 0
    scan = out.cpu().numpy().view(np.uint16)
    ref, _ = read_exr(os.path.join(GOLD, "egress", name + ".exr"))
    for k, c in enumerate(("B", "G", "R")):
        assert np.array_equal(scan[:, k, :], ref[c].view(np.uint16)), c
    a, b = str(tmp_path / "dev.exr"), str(tmp_path / "host.exr")
    write_exr_scanlines(a, scan)
    write_exr(b, img)
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_render_to_exr_scanlines_on_the_device(tmp_path):
    from helpers import golden_scene
    from take_amd import capi
    from take_amd.exr import write_exr_scanlines

    sd = golden_scene("cbox")
    sc = capi.Scene(sd)
    try:
        img = sc.render(spp=4, max_depth=5, seed=2)
        scan = sc.render_exr_scanlines(spp=4, max_depth=5, seed=2)
    finally:
        sc.close()
    want = float_to_half(img)  # (H, W, 3) R G B
    for k, c in enumerate((2, 1, 0)):
        assert np.array_equal(scan[:, k, :], want[:, :, c])
    a, b = str(tmp_path / "dev.exr"), str(tmp_path / "host.exr")
    write_exr_scanlines(a, scan)
    write_exr(b, img)
    assert open(a, "rb").read() == open(b, "rb").read()
