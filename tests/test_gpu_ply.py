"""PLY -> device mesh arrays on the GPU (take_hip_mesh_from_ply): the arrays the kernels write are
bit-identical to the `TriangleMesh` the reference's own parse_ply fills (tests/golden/ply: every encoding it reads,
identity / affine / projective to_world, normals through the reference's inverse), to the oracle's restatement on a
file too large to commit, and a scene built from a device-decoded mesh renders the same image, bit for bit, as the
scene built from host arrays — on the host SAH builder and on the device LBVH builder (which takes the positions where
the decode left them)."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ply as oply
from take_amd import capi, scenes
from take_amd import cdefs as D
from test_ply_cpu import CASES, assert_same_mesh, load_case, PLY

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_device_decode_is_bit_identical_to_the_reference_parser(name):
    data, xf, inv, ref = load_case(name)
    m = capi.DeviceMesh(data, material_id=3, to_world=xf, inv_to_world=inv)
    try:
        assert (m.n_vertices, m.n_faces) == (ref["positions"].shape[0], ref["indices"].shape[0])
        got = m.download()
        assert got.material_id == 3
        assert_same_mesh(got, ref)
    finally:
        m.close()
    f = capi.DeviceMesh(os.path.join(PLY, name + ".ply"), to_world=xf, inv_to_world=inv)  # (the memory-mapped file variant)
    try:
        assert_same_mesh(f.download(), ref)
    finally:
        f.close()


def big_ply(nv, nf, seed, bad_face=None, bad_index=None):
    rng = np.random.default_rng(seed)
    vert = np.zeros(nv, [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("u", "<f4"), ("v", "<f4")])
    for k in vert.dtype.names:
        vert[k] = rng.uniform(-1, 1, nv)
    face = np.zeros(nf, [("n", "u1"), ("i", "<i4", 3)])
    face["n"] = 3
    face["i"] = rng.integers(0, nv, (nf, 3))
    if bad_face is not None:
        face["n"][bad_face] = 4
    if bad_index is not None:
        face["i"][bad_index, 1] = nv
    hdr = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {nv}"] + [f"property float {k}" for k in vert.dtype.names] +
                    [f"element face {nf}", "property list uchar int vertex_indices", "end_header"]) + "\n"
    return hdr.encode() + vert.tobytes() + face.tobytes()


def test_million_face_file_matches_the_oracle_bit_for_bit():
    data = big_ply(600_001, 1_000_003, 5)  # (13-byte face rows: no row but the first is aligned)
    xf = np.array([[0.6, -0.8, 0.0, 1.0], [0.8, 0.6, 0.0, -2.0], [0.0, 0.0, 1.7, 0.5], [0.0, 0.0, 0.0, 1.0]])
    inv = np.linalg.inv(xf)
    want = oply.parse_ply(data, xf, inv)
    m = capi.DeviceMesh(data, to_world=xf, inv_to_world=inv)
    try:
        assert_same_mesh(m.download(), want)
    finally:
        m.close()


def test_faces_that_are_not_triangles_or_index_past_the_vertices_are_refused():
    with pytest.raises(capi.TakeError) as e:
        capi.DeviceMesh(big_ply(1000, 5000, 1, bad_face=4321))
    assert e.value.code == D.TAKE_E_INVALID and "not a triangle" in str(e.value)
    with pytest.raises(capi.TakeError) as e:
        capi.DeviceMesh(big_ply(1000, 5000, 1, bad_index=77))
    assert e.value.code == D.TAKE_E_INVALID and "past its vertex array" in str(e.value)
    # normals in the file and a to_world, but no inverse to push them through
    data = big_ply(10, 10, 1)
    xf = np.eye(4)
    out = D.TakeMesh()
    rc = capi.lib().take_hip_mesh_from_ply(data, len(data), xf.ctypes.data, None, 0, C.byref(out))
    assert rc == D.TAKE_E_INVALID and b"inverse(to_world)" in capi.lib().take_hip_last_error()


@pytest.mark.parametrize("nth", [1, 3, 5])
def test_failed_allocation_during_decode_is_reported(nth, monkeypatch):
    data = big_ply(1000, 5000, 2)
    monkeypatch.setenv("TAKE_HIP_FAIL_ALLOC", str(nth))
    with pytest.raises(capi.TakeError) as e:
        capi.DeviceMesh(data)
    assert e.value.code == D.TAKE_E_NOMEM
    monkeypatch.delenv("TAKE_HIP_FAIL_ALLOC")
    m = capi.DeviceMesh(data)
    assert m.n_faces == 5000
    m.close()


@pytest.mark.parametrize("builder,precision", [(D.TAKE_BUILDER_HOST_SAH, D.TAKE_PRECISION_F32), (D.TAKE_BUILDER_DEVICE_LBVH, D.TAKE_PRECISION_F32),
                                               (D.TAKE_BUILDER_HOST_SAH, D.TAKE_PRECISION_F64), (D.TAKE_BUILDER_AUTO, D.TAKE_PRECISION_MIXED)])
def test_scene_from_a_device_decoded_mesh_renders_the_same_image(builder, precision, tmp_path):
    """the soup of configs[1]'s shape, written as the binary PLY the reference's scenes use, decoded on the device and
    rendered, against the same scene from host arrays"""
    sd = scenes.soup_scene(20_000, 96, 64, spp=4)
    soup = max(range(len(sd.meshes)), key=lambda i: sd.meshes[i].indices.shape[0])
    host = sd.meshes[soup]
    pos32 = host.positions.astype(np.float32)  # (a PLY file of the reference's scenes stores floats)
    vert = np.zeros(len(pos32), [("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    vert["x"], vert["y"], vert["z"] = pos32[:, 0], pos32[:, 1], pos32[:, 2]
    face = np.zeros(len(host.indices), [("n", "u1"), ("i", "<i4", 3)])
    face["n"], face["i"] = 3, host.indices
    path = tmp_path / "soup.ply"
    hdr = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {len(vert)}", "property float x", "property float y",
                     "property float z", f"element face {len(face)}", "property list uchar int vertex_indices", "end_header"]) + "\n"
    path.write_bytes(hdr.encode() + vert.tobytes() + face.tobytes())
    sd_host = copy.copy(sd)
    sd_host.meshes = list(sd.meshes)
    sd_host.meshes[soup] = type(host)(pos32.astype(np.float64), host.indices, host.material_id, None, None)
    dm = capi.DeviceMesh(str(path), material_id=host.material_id)
    sd_dev = copy.copy(sd)
    sd_dev.meshes = list(sd.meshes)
    sd_dev.meshes[soup] = dm
    a = capi.Scene(sd_host, precision=precision, builder=builder)
    b = capi.Scene(sd_dev, precision=precision, builder=builder)
    try:
        assert a.stats() == b.stats()
        ia, ib = a.render(spp=4, max_depth=8, seed=3), b.render(spp=4, max_depth=8, seed=3)
        assert np.array_equal(ia, ib) and ia.mean() > 0.01
    finally:
        a.close(), b.close(), dm.close()


def mesh_to_ply(m):
    """a scene.Mesh as a double-precision PLY (values exact), normals / uvs if it has them"""
    cols = [("x", m.positions[:, 0]), ("y", m.positions[:, 1]), ("z", m.positions[:, 2])]
    if m.normals is not None:
        cols += [("nx", m.normals[:, 0]), ("ny", m.normals[:, 1]), ("nz", m.normals[:, 2])]
    if m.uvs is not None:
        cols += [("u", m.uvs[:, 0]), ("v", m.uvs[:, 1])]
    vert = np.zeros(len(m.positions), [(k, "<f8") for k, _ in cols])
    for k, c in cols:
        vert[k] = c
    face = np.zeros(len(m.indices), [("n", "u1"), ("i", "<u4", 3)])
    face["n"], face["i"] = 3, m.indices
    hdr = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {len(vert)}"] + [f"property double {k}" for k, _ in cols] +
                    [f"element face {len(face)}", "property list uchar uint vertex_indices", "end_header"]) + "\n"
    return hdr.encode() + vert.tobytes() + face.tobytes()


@pytest.mark.parametrize("name", ["meshlight", "mats"])
@pytest.mark.parametrize("builder,precision", [(D.TAKE_BUILDER_DEVICE_LBVH, D.TAKE_PRECISION_F32), (D.TAKE_BUILDER_HOST_SAH, D.TAKE_PRECISION_F64)])
def test_every_mesh_of_a_golden_scene_from_ply(name, builder, precision):
    """emissive meshes (their light records are made on the host from positions + normals), textured meshes (uvs),
    several meshes per scene: all of them decoded on the device; unit normals survive normalize() bit for bit only
    if |n| rounds to exactly 1, so the comparison scene takes the decoded arrays back"""
    from helpers import golden_scene

    sd = golden_scene(name)
    dms = [capi.DeviceMesh(mesh_to_ply(m), material_id=m.material_id) for m in sd.meshes]
    sd_host, sd_dev = copy.copy(sd), copy.copy(sd)
    sd_host.meshes = [dm.download() for dm in dms]
    sd_dev.meshes = list(dms)
    for h, m in zip(sd_host.meshes, sd.meshes):
        assert np.array_equal(h.positions, m.positions) and np.array_equal(h.indices, m.indices)
        assert (h.normals is None) == (m.normals is None) and (h.uvs is None) == (m.uvs is None)
        if m.normals is not None:
            assert np.abs(h.normals - m.normals).max() < 1e-15
    a = capi.Scene(sd_host, precision=precision, builder=builder)
    b = capi.Scene(sd_dev, precision=precision, builder=builder)
    try:
        ia, ib = a.render(spp=4, max_depth=6, seed=2), b.render(spp=4, max_depth=6, seed=2)
        assert np.array_equal(ia, ib) and ia.mean() > 0.01
    finally:
        a.close(), b.close()
        for dm in dms:
            dm.close()


def test_empty_elements_decode_to_empty_arrays():
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex {nv}\nproperty float x\nproperty float y\nproperty float z\n"
           "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
    m = capi.DeviceMesh(hdr.format(nv=0).encode())
    assert (m.n_vertices, m.n_faces) == (0, 0)
    got = m.download()
    assert got.positions.shape == (0, 3) and got.indices.shape == (0, 3)
    m.close()
    pts = np.arange(12, dtype="<f4")
    m = capi.DeviceMesh(hdr.format(nv=4).encode() + pts.tobytes())
    got = m.download()
    assert np.array_equal(got.positions, pts.reshape(4, 3).astype(np.float64)) and got.indices.shape == (0, 3)
    m.close()
    m.close()  # (idempotent)


def test_device_mesh_in_a_scene_group():
    """the shards of a group are replicas of the first scene: a device-decoded mesh goes in like a host one"""
    from helpers import golden_scene

    sd = golden_scene("meshlight")
    dms = [capi.DeviceMesh(mesh_to_ply(m), material_id=m.material_id) for m in sd.meshes]
    sd_dev = copy.copy(sd)
    sd_dev.meshes = list(dms)
    sd_host = copy.copy(sd)
    sd_host.meshes = [dm.download() for dm in dms]
    one = capi.Scene(sd_host, precision=D.TAKE_PRECISION_F32)
    grp = capi.SceneGroup(sd_dev, [0, 0, 0], precision=D.TAKE_PRECISION_F32)
    try:
        assert np.array_equal(grp.render(spp=4, max_depth=6, seed=9), one.render(spp=4, max_depth=6, seed=9))
    finally:
        one.close(), grp.close()
        for dm in dms:
            dm.close()
