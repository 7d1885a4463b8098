"""The drop-in, end to end: oracle/_ref/take_gpu is the reference's OWN main.cpp + XML/PLY parsers + imwrite, with
src/render.cpp replaced by take_amd/host/render_hip.cpp (the patch of INTEGRATION.md) and linked to
libtake_hip.so.  It is built in the authoring container by `make -C oracle gpu_cli` (reference sources stay where
they are) and travels to the GPU box as a binary.  Run on the golden XML scenes it must give exactly the image the
C ABI gives for the flattened scene, and leave the reference's `image.exr` behind."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLD, golden_scene
from take_amd import capi
from take_amd import cdefs as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_ref", "take_gpu")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0  # little endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)  # the reference writes Image3 order: row 0 = top


def run_cli(scene_xml, cwd, max_depth, env_extra):
    env = dict(os.environ, **env_extra)
    return subprocess.run([CLI, scene_xml, "-max_depth", str(max_depth)], cwd=cwd, env=env, capture_output=True,
                          text=True, timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "mats", "spherelight"])
def test_reference_cli_with_gpu_render_matches_c_abi(name, tmp_path):
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/take_gpu was not built (needs the reference sources: authoring container)")
    pfm = str(tmp_path / "out.pfm")
    r = run_cli(os.path.join(GOLD, "scenes", name + ".xml"), str(tmp_path), 5, {"TAKE_HIP_DUMP_PFM": pfm, "TAKE_HIP_SEED": "7"})
    assert r.returncode == 0, r.stderr[-2000:]
    for phrase in ("Parsing and constructing scene", "Building BVH...", "Rendering...", "Finish building rendering."):
        assert phrase in r.stdout  # the reference's own progress messages (src/render.cpp:25-29,48-50,58,83)
    exr = tmp_path / "image.exr"  # written by the reference's main.cpp:22 through its imwrite
    assert exr.exists() and exr.read_bytes()[:4] == bytes([0x76, 0x2F, 0x31, 0x01])
    got = read_pfm(pfm)
    sd = golden_scene(name)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    want = sc.render(spp=sd.spp, max_depth=5, seed=7)
    sc.close()
    assert got.shape == want.shape and np.array_equal(got, want)
    # egress: the EXR the reference's own writer left behind == the one the Python mirror writes for the same image
    from take_amd.exr import read_exr, write_exr

    write_exr(str(tmp_path / "mirror.exr"), want)
    ref_ch, ref_hdr = read_exr(str(exr))
    our_ch, our_hdr = read_exr(str(tmp_path / "mirror.exr"))
    for c in ("B", "G", "R"):
        assert np.array_equal(ref_ch[c].view(np.uint16), our_ch[c].view(np.uint16)), c
    assert ref_hdr["channels"] == our_hdr["channels"] and ref_hdr["compression"] == our_hdr["compression"]


def test_reference_cli_without_gpu_fails_like_the_parser_does():
    """no HIP device: the library has no CPU path; the error leaves as the reference's exception type"""
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/take_gpu was not built")
    try:
        capi.device_count()
        pytest.skip("a GPU is visible")
    except capi.TakeError:
        pass  # no HIP device: the case under test
    r = run_cli(os.path.join(GOLD, "scenes", "cbox.xml"), "/tmp", 5, {})
    assert r.returncode != 0
    assert "fl_exception" in r.stderr and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_python_main_writes_the_same_exr_as_the_reference_binary(tmp_path, monkeypatch):
    """`python -m take_amd.render scene.tkscene -max_depth 5` = the reference's main(): image.exr in the working
    directory, pixel for pixel the file the reference's writer produces for the same render"""
    from take_amd import render as R
    from take_amd.exr import read_exr

    monkeypatch.chdir(tmp_path)
    assert R.main([os.path.join(GOLD, "scenes", "cbox.tkscene"), "-t", "8", "-max_depth", "5"]) == 0
    ours, _ = read_exr(str(tmp_path / "image.exr"))
    if os.path.exists(CLI):
        sub = tmp_path / "ref"
        sub.mkdir()
        r = run_cli(os.path.join(GOLD, "scenes", "cbox.xml"), str(sub), 5, {"TAKE_HIP_SEED": "0"})
        assert r.returncode == 0, r.stderr[-2000:]
        ref, _ = read_exr(str(sub / "image.exr"))
        for c in ("B", "G", "R"):
            assert np.array_equal(ours[c].view(np.uint16), ref[c].view(np.uint16)), c
    assert R.render([]).shape == (0, 0, 3)  # src/render.cpp:10-12


@pytest.mark.gpu
def test_cli_reads_ply_on_the_device_and_leaves_big_endian_files_to_the_reference_parser(tmp_path):
    """take_gpu links take_amd/host/parse_ply_hip.cpp in place of the reference's parse_ply.cpp: little-endian files
    are decoded on the GPU; a big-endian file — same numbers, bytes swapped — goes to the reference's own parser
    (compiled in as parse_ply_host).  Both routes must fill the same TriangleMesh: same image, bit for bit.
    (ascii files are no test case: the reference's vendored tinyply throws "unexpected EOF" on an ascii face list.)"""
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/take_gpu was not built (needs the reference sources: authoring container)")
    import shutil

    from oracle import ply as oply

    scenes = os.path.join(GOLD, "scenes")
    images = {}
    for kind in ("little", "big"):
        d = tmp_path / kind
        d.mkdir()
        shutil.copy(os.path.join(scenes, "soup1k.xml"), d / "soup1k.xml")
        data = open(os.path.join(scenes, "soup1k.ply"), "rb").read()
        if kind == "big":
            elements, off = oply.read_header(data)
            (vn, vc, vp), (fn, fc, fp) = elements
            assert (vn, fn) == ("vertex", "face") and fp[0][1] == ("u1", "<i4")
            vert = np.frombuffer(data, np.dtype([(n, t) for n, t in vp]), vc, off)
            face = np.frombuffer(data, np.dtype([("n", "u1"), ("i", "<i4", 3)]), fc, off + vert.nbytes)
            data = (data[:off].replace(b"binary_little_endian", b"binary_big_endian") +
                    vert.astype([(n, ">f4") for n, _ in vp]).tobytes() + face.astype([("n", "u1"), ("i", ">i4", 3)]).tobytes())
        (d / "soup1k.ply").write_bytes(data)
        pfm = str(d / "out.pfm")
        r = run_cli(str(d / "soup1k.xml"), str(d), 5, {"TAKE_HIP_DUMP_PFM": pfm, "TAKE_HIP_SEED": "3"})
        assert r.returncode == 0, r.stderr[-2000:]
        images[kind] = read_pfm(pfm)
    assert np.array_equal(images["little"], images["big"]) and images["little"].mean() > 0.01
    sd = golden_scene("soup1k")
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    want = sc.render(spp=sd.spp, max_depth=5, seed=3)
    sc.close()
    assert np.array_equal(images["little"], want)


@pytest.mark.gpu
def test_cli_reads_serialized_meshes_on_the_device(tmp_path):
    """take_gpu links take_amd/host/parse_serialized_hip.cpp in place of the reference's parse_serialized.cpp: the soup of
    the golden scene, rewritten as sub-mesh 1 of a Mitsuba-serialized file (version 4, float), must give the image of the
    PLY scene bit for bit — same vertices, same faces, through the other loader"""
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/take_gpu was not built (needs the reference sources: authoring container)")
    import struct
    import zlib

    from oracle import ply as oply

    scenes = os.path.join(GOLD, "scenes")
    mesh = oply.parse_ply(open(os.path.join(scenes, "soup1k.ply"), "rb").read())

    def sub(pos, faces, name):
        body = struct.pack("<I", 0x1000) + name + b"\0" + struct.pack("<QQ", len(pos), len(faces))
        return struct.pack("<HH", 0x041C, 4) + zlib.compress(body + pos.astype("<f4").tobytes() + faces.astype("<i4").tobytes())

    decoy = sub(np.zeros((3, 3)), np.array([[0, 1, 2]]), b"decoy")
    soup = sub(mesh["positions"], mesh["indices"], b"soup")
    data = decoy + soup + struct.pack("<QQI", 0, len(decoy), 2)
    (tmp_path / "soup1k.serialized").write_bytes(data)
    xml = open(os.path.join(scenes, "soup1k.xml")).read()
    old = '<shape type="ply"><string name="filename" value="soup1k.ply"/>'
    assert xml.count(old) == 1
    xml = xml.replace(old, '<shape type="serialized"><string name="filename" value="soup1k.serialized"/><integer name="shapeIndex" value="1"/>')
    (tmp_path / "soup1k.xml").write_text(xml)
    pfm = str(tmp_path / "out.pfm")
    r = run_cli(str(tmp_path / "soup1k.xml"), str(tmp_path), 5, {"TAKE_HIP_DUMP_PFM": pfm, "TAKE_HIP_SEED": "3"})
    assert r.returncode == 0, r.stderr[-2000:]
    sd = golden_scene("soup1k")
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    want = sc.render(spp=sd.spp, max_depth=5, seed=3)
    sc.close()
    assert np.array_equal(read_pfm(pfm), want)
