"""Mitsuba-serialized meshes ("direct PLY/serialized -> device buffers").  The reference's loader
(src/parse/parse_serialized.cpp:174-256) pulls the zlib stream through ZStream::read three scalars per vertex; the
library inflates it in one pass on the host and decodes the blocks on the device (take_hip_mesh_from_serialized).
CPU: the oracle's numpy restatement (oracle/serialized.py) against the arrays the reference's OWN parser made of the
committed files (tests/golden/serialized, written by `oracle/gen_golden.py serialized` through oracle/_ref).
GPU: the device arrays bit-identical to the same golden arrays, and to the oracle on a file too large to commit."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import GOLD
from oracle import serialized as oser
from take_amd import capi
from take_amd import cdefs as D
from test_ply_cpu import assert_same_mesh

DIR = os.path.join(GOLD, "serialized")
CASES = sorted(f[:-len("_mesh.f64")] for f in os.listdir(DIR) if f.endswith("_mesh.f64"))


def load_case(key):
    name, idx = key.rsplit("_", 1)
    data = open(os.path.join(DIR, name + ".serialized"), "rb").read()
    xf = np.fromfile(os.path.join(DIR, key + "_xform.f64"), "<f8").reshape(4, 4)
    a = np.fromfile(os.path.join(DIR, key + "_mesh.f64"), "<f8")
    nv, nf, has_n, has_uv = (int(x) for x in a[:4])
    inv = a[4:20].reshape(4, 4)
    o = 20
    ref = {"positions": a[o:o + 3 * nv].reshape(nv, 3)}
    o += 3 * nv
    ref["indices"] = a[o:o + 3 * nf].reshape(nf, 3).astype(np.int32)
    o += 3 * nf
    ref["normals"] = a[o:o + 3 * nv].reshape(nv, 3) if has_n else None
    o += 3 * nv * has_n
    ref["uvs"] = a[o:o + 2 * nv].reshape(nv, 2) if has_uv else None
    o += 2 * nv * has_uv
    assert o == a.size
    return data, int(idx), xf, inv, ref


def test_fixture_set_covers_both_versions_and_every_block():
    assert len(CASES) == 5 and any(c.startswith("v3_") for c in CASES) and any(c.startswith("v4_") for c in CASES)


@pytest.mark.parametrize("key", CASES)
def test_serialized_oracle_matches_reference(key):
    data, idx, xf, inv, ref = load_case(key)
    assert_same_mesh(oser.parse_serialized(data, idx, xf, inv), ref)


def test_decode_without_gpu_is_an_error_not_a_host_parse():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-GPU contract is checked in the CPU container")
    data = load_case(CASES[0])[0]
    m = D.TakeMesh()
    rc = capi.lib().take_hip_mesh_from_serialized(data, len(data), 0, None, None, 0, C.byref(m))
    assert rc == D.TAKE_E_NO_GPU and not m.positions


def blob(version, flags, nv, nf, seed, truncate=0):
    rng = np.random.default_rng(seed)
    t = "<f8" if flags & 0x2000 else "<f4"
    body = struct.pack("<I", flags) + (b"big\0" if version == 4 else b"") + struct.pack("<QQ", nv, nf)
    body += rng.uniform(-1, 1, (nv, 3)).astype(t).tobytes()
    if flags & 1:
        body += rng.normal(size=(nv, 3)).astype(t).tobytes()
    if flags & 2:
        body += rng.uniform(0, 1, (nv, 2)).astype(t).tobytes()
    if flags & 8:
        body += rng.uniform(0, 1, (nv, 3)).astype(t).tobytes()
    body += rng.integers(0, nv, (nf, 3)).astype("<i4").tobytes()
    if truncate:
        body = body[:-truncate]
    return struct.pack("<HH", 0x041C, version) + zlib.compress(body, 1) + struct.pack("<QI" if version == 4 else "<II", 0, 1)


@pytest.mark.parametrize("data,idx,msg", [
    (struct.pack("<HH", 0x041C, 7) + b"x" * 40, 0, "unknown format version"),
    (blob(4, 0x1000, 10, 10, 1), 1, "shape index 1 of 1"),
    (blob(4, 0x1000, 10, 10, 1), -1, "negative shape index"),
    (struct.pack("<HH", 0x041C, 4) + b"this is not a zlib stream at all, is it?" * 2, 0, "inflate()"),
    (b"\x1c\x04", 0, "shorter than its header"),
])
def test_malformed_files_are_refused_before_any_device_work(data, idx, msg):
    m = D.TakeMesh()
    rc = capi.lib().take_hip_mesh_from_serialized(data, len(data), idx, None, None, 0, C.byref(m))
    assert rc == D.TAKE_E_INVALID and msg.encode() in capi.lib().take_hip_last_error(), capi.lib().take_hip_last_error()


# ------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_device_decode_is_bit_identical_to_the_reference_parser(key, tmp_path):
    data, idx, xf, inv, ref = load_case(key)
    m = capi.DeviceMesh(data, material_id=2, to_world=xf, inv_to_world=inv, shape_index=idx)
    try:
        assert_same_mesh(m.download(), ref)
    finally:
        m.close()
    p = tmp_path / "m.serialized"
    p.write_bytes(data)
    f = capi.DeviceMesh(str(p), to_world=xf, inv_to_world=inv, shape_index=idx)  # (the memory-mapped file variant)
    try:
        assert_same_mesh(f.download(), ref)
    finally:
        f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("version,flags", [(4, 0x1000 | 1 | 2 | 8), (3, 0x2000 | 1)])
def test_million_face_stream_matches_the_oracle_bit_for_bit(version, flags):
    data = blob(version, flags, 500_003, 1_000_001, 11)
    xf = np.array([[0.6, -0.8, 0.0, 1.0], [0.8, 0.6, 0.0, -2.0], [0.0, 0.0, 1.7, 0.5], [0.0, 0.0, 0.0, 1.0]])
    inv = np.linalg.inv(xf)
    want = oser.parse_serialized(data, 0, xf, inv)
    m = capi.DeviceMesh(data, to_world=xf, inv_to_world=inv)
    try:
        assert_same_mesh(m.download(), want)
    finally:
        m.close()


@pytest.mark.gpu
def test_truncated_stream_and_bad_indices_are_refused():
    with pytest.raises(capi.TakeError) as e:
        capi.DeviceMesh(blob(4, 0x1000, 1000, 5000, 3, truncate=100))
    assert e.value.code == D.TAKE_E_INVALID and "past the end of the stream" in str(e.value)
    bad = bytearray(struct.pack("<I", 0x1000) + b"\0" + struct.pack("<QQ", 3, 1) + np.zeros(9, "<f4").tobytes() + np.array([0, 1, 3], "<i4").tobytes())
    data = struct.pack("<HH", 0x041C, 4) + zlib.compress(bytes(bad))
    with pytest.raises(capi.TakeError) as e:
        capi.DeviceMesh(data)
    assert "past its vertex array" in str(e.value)
