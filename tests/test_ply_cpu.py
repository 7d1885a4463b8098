"""PLY ingestion, CPU side: the oracle's numpy restatement of the reference's parse_ply
(oracle/ply.py) against the arrays the reference's OWN parser made of the committed files (tests/golden/ply, written by
`oracle/gen_golden.py ply` through oracle/_ref/ref_harness), and the library's header parser — the only part of the
device decode that runs on the host — on the same files and on the encodings it must refuse."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import GOLD
from oracle import ply as oply
from take_amd import capi
from take_amd import cdefs as D

PLY = os.path.join(GOLD, "ply")
CASES = sorted(f[:-4] for f in os.listdir(PLY) if f.endswith(".ply"))


def load_case(name):
    """-> (file bytes, to_world, the reference's inverse(to_world), the reference's TriangleMesh arrays)"""
    data = open(os.path.join(PLY, name + ".ply"), "rb").read()
    xf = np.fromfile(os.path.join(PLY, name + "_xform.f64"), "<f8").reshape(4, 4)
    a = np.fromfile(os.path.join(PLY, name + "_mesh.f64"), "<f8")
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
    return data, xf, inv, ref


def assert_same_mesh(got, ref):
    """bit-exact: integer indices, and doubles compared as bit patterns (a signed zero is a difference too)"""
    for k in ("positions", "indices", "normals", "uvs"):
        g = got[k] if isinstance(got, dict) else getattr(got, k)
        if ref[k] is None:
            assert g is None, k
            continue
        assert g is not None and g.shape == ref[k].shape, k
        if k == "indices":
            assert np.array_equal(g, ref[k]), k
        else:
            assert np.array_equal(np.ascontiguousarray(g).view(np.uint64), np.ascontiguousarray(ref[k]).view(np.uint64)), k


def test_fixture_set_covers_the_encodings():
    assert {"f32_plain", "f32_normals_uvs_affine", "f64_all_projective", "u16_faces_first", "i8_indices", "u8_i16"} <= set(CASES)


@pytest.mark.parametrize("name", CASES)
def test_ply_oracle_matches_reference(name):
    data, xf, inv, ref = load_case(name)
    assert_same_mesh(oply.parse_ply(data, xf, inv), ref)


@pytest.mark.parametrize("name", CASES)
def test_library_reads_the_header_like_the_reference(name):
    data, _, _, ref = load_case(name)
    lay = capi.ply_layout(data)
    assert lay["n_vertices"] == ref["positions"].shape[0] and lay["n_faces"] == ref["indices"].shape[0]
    assert bool(lay["has_normals"]) == (ref["normals"] is not None) and bool(lay["has_uvs"]) == (ref["uvs"] is not None)
    elements, hdr = oply.read_header(data)
    assert lay["header_bytes"] == hdr
    # element offsets and strides: the file is header + the rows of the elements in header order
    sizes = {}
    off = hdr
    for ename, count, props in elements:
        stride = sum(np.dtype(t).itemsize if not isinstance(t, tuple) else np.dtype(t[0]).itemsize + 3 * np.dtype(t[1]).itemsize
                     for _, t in props)
        sizes[ename] = (off, stride)
        off += count * stride
    assert off == len(data)
    assert (lay["vertex_offset"], lay["vertex_stride"]) == sizes["vertex"]
    assert (lay["face_offset"], lay["face_stride"]) == sizes["face"]


def header(*lines):
    return ("\n".join(("ply",) + lines + ("end_header",)) + "\n").encode()


V = ("element vertex 1", "property float x", "property float y", "property float z")
F = ("element face 1", "property list uchar int vertex_indices")
BODY = b"\0" * 12 + b"\3" + b"\0" * 12


@pytest.mark.parametrize("data,msg", [
    (header("format ascii 1.0", *V, *F) + b"0 0 0\n3 0 0 0\n", "unsupported PLY encoding"),
    (header("format binary_big_endian 1.0", *V, *F) + BODY, "unsupported PLY encoding"),
    (header("format binary_little_endian 1.0", "element vertex 1", "property float x", "property float y", *F) + BODY, "positions not found"),
    (header("format binary_little_endian 1.0", *V) + BODY, "indices not found"),
    (header("format binary_little_endian 1.0", *V, "element face 1", "property list uchar int vertex_index") + BODY, "indices not found"),
    (header("format binary_little_endian 1.0", *V, "property list uchar float weights", *F) + BODY, "unsupported: list property"),
    (header("format binary_little_endian 1.0", "element vertex 1", "property int x", "property int y", "property int z", *F) + BODY,
     "neither float nor double"),
    (header("format binary_little_endian 1.0", "element vertex 1", "property float x", "property double y", "property float z", *F) + BODY,
     "different types"),
    (header("format binary_little_endian 1.0", "element strip 2", "property list uchar int ids", *V, *F) + BODY, "ahead of the mesh data"),
    (header("format binary_little_endian 1.0", *V, *F) + BODY[:-1], "shorter than its header says"),
    (b"plx\nformat binary_little_endian 1.0\nend_header\n", "not a PLY file"),
    (b"ply\nformat binary_little_endian 1.0\n" + b"x" * 100, "no end_header"),
    (header("format binary_little_endian 1.0", "element vertex 1", "property quaternion x", *F) + BODY, "unknown property type"),
])
def test_header_errors(data, msg):
    with pytest.raises(capi.TakeError) as e:
        capi.ply_layout(data)
    assert e.value.code == D.TAKE_E_INVALID and msg in str(e.value), str(e.value)


def test_header_tolerates_crlf_comments_and_trailing_elements():
    h = ("ply\r\nformat binary_little_endian 1.0\r\ncomment hello world\r\nobj_info x\r\n" + "\r\n".join(V + F) +
         "\r\nelement edge 2\r\nproperty list uchar int e\r\nend_header\r\n").encode()
    lay = capi.ply_layout(h + BODY)
    assert lay["header_bytes"] == len(h) and lay["vertex_stride"] == 12 and lay["face_stride"] == 13
    assert lay["vertex_offset"] == len(h) and lay["face_offset"] == len(h) + 12


def test_decode_without_gpu_is_an_error_not_a_host_parse():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-GPU contract is checked in the CPU container")
    data = load_case("f32_plain")[0]
    m = D.TakeMesh()
    rc = capi.lib().take_hip_mesh_from_ply(data, len(data), None, None, 0, C.byref(m))
    assert rc == D.TAKE_E_NO_GPU and not m.positions
