#!/usr/bin/env python3
"""gen_golden.py — TEST INFRASTRUCTURE: generates tests/golden/ by running the compiled reference.

Runs only in the authoring container (needs oracle/_ref/ref_harness = TaKe built from
/root/reference by `make -C oracle ref`).  Everything it writes is *data*: scene inputs authored
here in the reference's XML dialect (+ binary PLY / TGA files), the flattened `.tkscene` replay
of what the reference's parser made of them, and the reference's outputs on those inputs.

Layout of tests/golden/:
  scenes/<name>.xml (+ .ply/.tga)     inputs, authored here
  scenes/<name>.tkscene               reference parse_scene() -> take_flatten.hpp -> take_sceneio.hpp
  render/<name>_d<D>.f64              reference render() (seed-patched per tile), float64 [2 + H*W*3]
  tables/<fn>_in.f64, <fn>_out.f64    per-function known-answer tables (column layouts below and in
                                      oracle/ref_harness.cpp)
  manifest.json                       what was generated, with shapes

The fixtures pin the g++/libstdc++ behaviour of the reference.
"""
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLD, "scenes")
RENDER = os.path.join(GOLD, "render")
TABLES = os.path.join(GOLD, "tables")


def run(*args):
    r = subprocess.run([HARNESS, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        raise SystemExit(f"ref_harness {' '.join(map(str, args))} failed ({r.returncode})")
    return r.stdout


# ----------------------------------------------------------------------------- file writers
def write_ply(path, pos, faces, normals=None, uvs=None):
    """binary little-endian PLY as src/parse/parse_ply.cpp:9-123 reads it"""
    pos = np.asarray(pos, np.float32)
    faces = np.asarray(faces, np.int32)
    props = ["property float x", "property float y", "property float z"]
    cols = [pos]
    if normals is not None:
        props += ["property float nx", "property float ny", "property float nz"]
        cols.append(np.asarray(normals, np.float32))
    if uvs is not None:
        props += ["property float u", "property float v"]
        cols.append(np.asarray(uvs, np.float32))
    vert = np.concatenate(cols, axis=1).astype("<f4")
    hdr = "\n".join(
        ["ply", "format binary_little_endian 1.0", f"element vertex {len(pos)}", *props,
         f"element face {len(faces)}", "property list uchar int vertex_indices", "end_header"]) + "\n"
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(vert.tobytes())
        rec = np.zeros(len(faces), dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"] = 3
        rec["i"] = faces
        f.write(rec.tobytes())


def write_tga(path, rgb8):
    """uncompressed 24-bit TGA, top-left origin (stb_image reads it; imread3 src/image.cpp:80-108)"""
    h, w, _ = rgb8.shape
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 24, 0x20)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(rgb8[:, :, ::-1].astype(np.uint8).tobytes())


def soup(n, seed, half=0.9, jitter=0.02):
    """random triangle soup: centres ~U(-half,half)^3, vertex offsets ~U(-jitter,jitter)^3"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-half, half, (n, 1, 3))
    v = c + rng.uniform(-jitter, jitter, (n, 3, 3))
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.int32).reshape(n, 3)


# ----------------------------------------------------------------------------- scenes
HEAD = """<scene version="0.5.0">
  <sensor type="perspective"><float name="fov" value="39"/>
    <transform name="toWorld"><lookat origin="0,0,3.9" target="0,0,0" up="0,1,0"/></transform>
    <sampler type="independent"><integer name="sampleCount" value="{spp}"/></sampler>
    <film type="hdrfilm"><integer name="width" value="{w}"/><integer name="height" value="{h}"/></film></sensor>
"""
WALLS = """  <bsdf type="diffuse" id="white"><rgb name="reflectance" value="0.73 0.73 0.73"/></bsdf>
  <bsdf type="diffuse" id="red"><rgb name="reflectance" value="0.65 0.05 0.05"/></bsdf>
  <bsdf type="diffuse" id="green"><rgb name="reflectance" value="0.12 0.45 0.15"/></bsdf>
  <shape type="rectangle"><transform name="toWorld"><translate z="-1"/></transform><ref id="{back}"/></shape>
  <shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="-90"/><translate y="-1"/></transform><ref id="{floor}"/></shape>
  <shape type="rectangle"><transform name="toWorld"><rotate x="1" angle="90"/><translate y="1"/></transform><ref id="white"/></shape>
  <shape type="rectangle"><transform name="toWorld"><rotate y="1" angle="90"/><translate x="-1"/></transform><ref id="red"/></shape>
  <shape type="rectangle"><transform name="toWorld"><rotate y="1" angle="-90"/><translate x="1"/></transform><ref id="green"/></shape>
"""
QUADLIGHT = """  <shape type="rectangle"><transform name="toWorld"><scale value="0.3"/><rotate x="1" angle="90"/><translate y="0.99"/></transform>
    <ref id="white"/><emitter type="area"><rgb name="radiance" value="17 12 4"/></emitter></shape>
"""


def scene_cbox(w=64, h=64, spp=8):
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <background><rgb name="radiance" value="0 0 0"/></background>\n'
            + '  <bsdf type="plastic" id="plas"><rgb name="reflectance" value="0.2 0.3 0.8"/><float name="ior" value="1.5"/></bsdf>\n'
            + '  <bsdf type="blinn_microfacet" id="mf"><rgb name="reflectance" value="0.8 0.7 0.3"/><float name="exponent" value="50"/></bsdf>\n'
            + WALLS.format(back="white", floor="white") + QUADLIGHT
            + '  <shape type="sphere"><point name="center" x="-0.4" y="-0.6" z="-0.3"/><float name="radius" value="0.4"/><ref id="plas"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.45" y="-0.65" z="0.3"/><float name="radius" value="0.35"/><ref id="mf"/></shape>\n'
            + "</scene>\n")


def scene_mats(w=64, h=48, spp=8):
    # every material with a real implementation, a bitmap texture with uvscale, a point light,
    # default background (0.5 grey: no <background>), non-square film
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <texture type="bitmap" id="checker"><string name="filename" value="tex8x4.tga"/><float name="uvscale" value="2"/><float name="uoffset" value="0.25"/></texture>\n'
            + '  <bsdf type="diffuse" id="textured"><ref name="reflectance" id="checker"/></bsdf>\n'
            + '  <bsdf type="disneydiffuse" id="dd"><rgb name="baseColor" value="0.6 0.5 0.4"/><float name="roughness" value="0.6"/><float name="subsurface" value="0.3"/></bsdf>\n'
            + '  <bsdf type="mirror" id="mir"><rgb name="reflectance" value="0.9 0.85 0.7"/></bsdf>\n'
            + '  <bsdf type="phong" id="ph"><rgb name="reflectance" value="0.5 0.6 0.3"/><float name="exponent" value="20"/></bsdf>\n'
            + '  <bsdf type="blinn" id="bl"><rgb name="reflectance" value="0.3 0.4 0.7"/><float name="exponent" value="40"/></bsdf>\n'
            + '  <bsdf type="twosided" id="two"><bsdf type="disneymetal"><rgb name="baseColor" value="0.7 0.2 0.2"/></bsdf></bsdf>\n'
            + '  <bsdf type="disneysheen" id="sh"><rgb name="baseColor" value="0.2 0.7 0.6"/></bsdf>\n'
            + '  <emitter type="point"><point name="position" x="0" y="0.5" z="0"/><rgb name="intensity" value="5 5 5"/></emitter>\n'
            + WALLS.format(back="textured", floor="dd") + QUADLIGHT
            + '  <shape type="sphere"><point name="center" x="-0.55" y="-0.7" z="-0.4"/><float name="radius" value="0.3"/><ref id="mir"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.0" y="-0.7" z="-0.1"/><float name="radius" value="0.3"/><ref id="ph"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.55" y="-0.7" z="0.2"/><float name="radius" value="0.3"/><ref id="bl"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="-0.5" y="0.1" z="-0.5"/><float name="radius" value="0.25"/><ref id="two"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.5" y="0.1" z="-0.5"/><float name="radius" value="0.25"/><ref id="sh"/></shape>\n'
            + "</scene>\n")


def scene_soup(w=64, h=64, spp=8):
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <background><rgb name="radiance" value="0 0 0"/></background>\n'
            + WALLS.format(back="white", floor="white") + QUADLIGHT
            + '  <shape type="ply"><string name="filename" value="soup1k.ply"/><boolean name="faceNormals" value="true"/><ref id="white"/></shape>\n'
            + "</scene>\n")


def scene_spherelight(w=48, h=64, spp=8):
    # sphere area light (cone sampling, src/shape.cpp:125-144), coloured background seen through the open front
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <background><rgb name="radiance" value="0.2 0.3 0.4"/></background>\n'
            + WALLS.format(back="white", floor="white")
            + '  <shape type="sphere"><point name="center" x="0.2" y="0.45" z="0.1"/><float name="radius" value="0.2"/><ref id="white"/>'
            + '<emitter type="area"><rgb name="radiance" value="9 9 12"/></emitter></shape>\n'
            + '  <shape type="sphere"><point name="center" x="-0.35" y="-0.6" z="-0.1"/><float name="radius" value="0.4"/><ref id="green"/></shape>\n'
            + "</scene>\n")


def scene_meshlight(w=64, h=64, spp=8):
    # emissive PLY mesh with computed vertex normals (compute_normals.cpp), a PLY mesh carrying uvs + normals
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <background><rgb name="radiance" value="0.05 0.05 0.05"/></background>\n'
            + '  <texture type="bitmap" id="checker"><string name="filename" value="tex8x4.tga"/></texture>\n'
            + '  <bsdf type="diffuse" id="textured"><ref name="reflectance" id="checker"/></bsdf>\n'
            + WALLS.format(back="white", floor="white")
            + '  <shape type="ply"><string name="filename" value="tetra_light.ply"/><ref id="white"/>'
            + '<emitter type="area"><rgb name="radiance" value="12 10 8"/></emitter></shape>\n'
            + '  <shape type="ply"><string name="filename" value="wavy_uv.ply"/><ref id="textured"/></shape>\n'
            + "</scene>\n")


def scene_disney(w=64, h=64, spp=8):
    # every Disney alternative of the reference's Material with explicit parameters (src/parse/parse_scene.cpp:562-700):
    # upstream renders Lambert clones of them (src/materials/disney_*.inl) — that is what the render golden of this
    # scene pins; the flattened parameters feed the Burley lobes (tags 12..16, TakeBuildOpts.burley_lobes)
    return (HEAD.format(spp=spp, w=w, h=h)
            + '  <background><rgb name="radiance" value="0 0 0"/></background>\n'
            + '  <bsdf type="disneymetal" id="metal"><rgb name="baseColor" value="0.95 0.64 0.54"/><float name="roughness" value="0.35"/><float name="anisotropic" value="0.7"/></bsdf>\n'
            + '  <bsdf type="disneyglass" id="glass"><rgb name="baseColor" value="0.95 0.97 1.0"/><float name="roughness" value="0.15"/><float name="eta" value="1.5"/></bsdf>\n'
            + '  <bsdf type="disneyglass" id="glass2"><rgb name="baseColor" value="1.0 0.85 0.7"/><float name="roughness" value="0.3"/><float name="anisotropic" value="0.4"/><float name="eta" value="1.33"/></bsdf>\n'
            + '  <bsdf type="disneyclearcoat" id="coat"><float name="clearcoatGloss" value="0.6"/></bsdf>\n'
            + '  <bsdf type="disneysheen" id="sheen"><rgb name="baseColor" value="0.3 0.5 0.9"/><float name="sheenTint" value="0.7"/></bsdf>\n'
            + '  <bsdf type="disneybsdf" id="pr1"><rgb name="baseColor" value="0.8 0.3 0.2"/><float name="specularTransmission" value="0.6"/>'
            + '<float name="roughness" value="0.25"/><float name="clearcoat" value="0.8"/><float name="clearcoatGloss" value="0.5"/>'
            + '<float name="sheen" value="0.4"/><float name="eta" value="1.45"/></bsdf>\n'
            + '  <bsdf type="principled" id="pr2"><rgb name="baseColor" value="0.9 0.75 0.3"/><float name="metallic" value="0.8"/>'
            + '<float name="roughness" value="0.4"/><float name="anisotropic" value="0.6"/><float name="specularTint" value="0.5"/>'
            + '<float name="subsurface" value="0.3"/></bsdf>\n'
            + WALLS.format(back="white", floor="white") + QUADLIGHT
            + '  <shape type="sphere"><point name="center" x="-0.6" y="-0.69" z="-0.3"/><float name="radius" value="0.3"/><ref id="metal"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.1" y="-0.67" z="0.35"/><float name="radius" value="0.32"/><ref id="glass"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.65" y="-0.74" z="-0.35"/><float name="radius" value="0.25"/><ref id="sheen"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="-0.55" y="0.0" z="-0.55"/><float name="radius" value="0.27"/><ref id="pr1"/></shape>\n'
            + '  <shape type="sphere"><point name="center" x="0.55" y="0.05" z="-0.5"/><float name="radius" value="0.27"/><ref id="pr2"/></shape>\n'
            # no shape refers to "coat": upstream's DisneyClearcoat eval returns an uninitialised vector
            # (disney_clearcoat.inl:26), so a render with it cannot be a golden; the material is declared for the
            # flattened parameter
            + '  <shape type="ply"><string name="filename" value="cube.ply"/><boolean name="faceNormals" value="true"/><ref id="glass2"/></shape>\n'
            + "</scene>\n")


def make_disney():
    """`python oracle/gen_golden.py disney`: scenes/disney.xml (+ cube.ply), its flattened .tkscene and the reference's
    own render of it, added to the manifest without touching the other fixtures."""
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    c = np.array([-0.15, -0.79, -0.5])
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) * 0.2 + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    faces = [t for a, b, cc, d in quads for t in ((a, b, cc), (a, cc, d))]
    write_ply(os.path.join(SCENES, "cube.ply"), v, np.array(faces, np.int32))
    xml = os.path.join(SCENES, "disney.xml")
    with open(xml, "w") as f:
        f.write(scene_disney())
    run("flatten", xml, os.path.join(SCENES, "disney.tkscene"))
    os.makedirs(RENDER, exist_ok=True)
    out = os.path.join(RENDER, "disney_d8.f64")
    run("render", xml, 8, 4, out)
    a = np.fromfile(out, "<f8")
    man["render/disney_d8"] = {"w": int(a[0]), "h": int(a[1]), "mean": float(a[2:].mean())}
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


def make_scene_files():
    os.makedirs(SCENES, exist_ok=True)
    # 8x4 texture
    yy, xx = np.mgrid[0:4, 0:8]
    tex = np.stack([(xx * 32 + 16) % 256, (yy * 64 + 31) % 256, ((xx + yy) % 2) * 200 + 30], axis=-1).astype(np.uint8)
    write_tga(os.path.join(SCENES, "tex8x4.tga"), tex)
    p, f = soup(1000, 1234, half=0.85, jitter=0.06)
    write_ply(os.path.join(SCENES, "soup1k.ply"), p, f)
    # tetrahedron light near the ceiling
    tp = np.array([[0, 0.9, 0], [-0.3, 0.6, -0.2], [0.3, 0.6, -0.2], [0, 0.6, 0.3]], np.float32)
    tf = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32)
    write_ply(os.path.join(SCENES, "tetra_light.ply"), tp, tf)
    # wavy grid with uvs and analytic normals
    n = 9
    gx, gz = np.meshgrid(np.linspace(-0.8, 0.8, n), np.linspace(-0.6, 0.8, n), indexing="xy")
    gy = -0.7 + 0.12 * np.sin(3 * gx) * np.cos(2.5 * gz)
    pos = np.stack([gx, gy, gz], -1).reshape(-1, 3)
    dydx = 0.36 * np.cos(3 * gx) * np.cos(2.5 * gz)
    dydz = -0.3 * np.sin(3 * gx) * np.sin(2.5 * gz)
    nrm = np.stack([-dydx, np.ones_like(gx), -dydz], -1).reshape(-1, 3)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    uv = np.stack([(gx + 0.8) / 1.6, (gz + 0.6) / 1.4], -1).reshape(-1, 2)
    faces = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i
            faces += [[a, a + n, a + 1], [a + 1, a + n, a + n + 1]]
    write_ply(os.path.join(SCENES, "wavy_uv.ply"), pos, np.array(faces, np.int32), normals=nrm, uvs=uv)
    scenes = {"cbox": scene_cbox(), "mats": scene_mats(), "soup1k": scene_soup(),
              "spherelight": scene_spherelight(), "meshlight": scene_meshlight()}
    for name, xml in scenes.items():
        with open(os.path.join(SCENES, name + ".xml"), "w") as f:
            f.write(xml)
    return list(scenes)


# ----------------------------------------------------------------------------- tables
def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def table(name, arr, *pre):
    os.makedirs(TABLES, exist_ok=True)
    fin = os.path.join(TABLES, f"{name}_in.f64")
    fout = os.path.join(TABLES, f"{name}_out.f64")
    np.ascontiguousarray(arr, dtype="<f8").tofile(fin)
    cmd = name.split("-")[0]
    run(cmd, *pre, fin, fout)
    out = np.fromfile(fout, "<f8")
    return {"in": list(np.shape(arr)), "out_len": int(out.size)}


def make_tables(man):
    rng = np.random.default_rng(20261004)
    N = 1024
    # random_real: first 64 draws of seeds 0..7  (src/take.h:89-91)
    man["random_real"] = table("random_real", np.arange(8, dtype=np.float64))
    # slab test (src/bbox.h:18-32), incl. zero direction components, degenerate (flat) boxes, origin inside
    c = rng.uniform(-1, 1, (N, 3))
    e = rng.uniform(0, 0.5, (N, 3))
    e[::7, 0] = 0.0                                  # flat boxes
    o = rng.uniform(-2, 2, (N, 3))
    o[::5] = c[::5] + rng.uniform(-0.1, 0.1, (len(c[::5]), 3)) * e[::5]   # origin inside
    d = unit(rng.normal(size=(N, 3)))
    d[::11, 1] = 0.0                                 # axis-parallel
    d[::13, 2] = -0.0
    tmin = np.full((N, 1), 1e-7)
    tmax = np.where(rng.uniform(size=(N, 1)) < 0.5, np.inf, rng.uniform(0.1, 4, (N, 1)))
    man["slab"] = table("slab", np.hstack([c - e, c + e, o, d, tmin, tmax]))
    # triangle (src/shape.cpp:44-110): with/without normals and uvs, edge-on rays, bounded tmax
    v0 = rng.uniform(-1, 1, (N, 3)); v1 = v0 + rng.uniform(-0.6, 0.6, (N, 3)); v2 = v0 + rng.uniform(-0.6, 0.6, (N, 3))
    bary = rng.dirichlet([1, 1, 1], N)
    bary[::9] = rng.uniform(-0.3, 1.3, (len(bary[::9]), 3))          # misses
    tgt = bary[:, :1] * v0 + bary[:, 1:2] * v1 + bary[:, 2:3] * v2
    o = rng.uniform(-2.5, 2.5, (N, 3))
    d = unit(tgt - o)
    d[::17] = unit((v1 - v0)[::17] + 1e-9)                          # parallel to the plane
    dist = np.linalg.norm(tgt - o, axis=1, keepdims=True)
    tmax = np.where(rng.uniform(size=(N, 1)) < 0.3, dist * rng.uniform(0.5, 1.5, (N, 1)), np.inf)
    flags = rng.integers(0, 2, (N, 2)).astype(np.float64)
    nn = unit(rng.normal(size=(N, 9)).reshape(N, 3, 3)).reshape(N, 9)
    uv = rng.uniform(0, 1, (N, 6))
    man["tri"] = table("tri", np.hstack([v0, v1, v2, o, d, np.full((N, 1), 1e-7), tmax, flags, nn, uv]))
    # sphere (src/shape.cpp:13-42): outside, inside, tangent-ish, bounded
    c = rng.uniform(-1, 1, (N, 3)); r = rng.uniform(0.05, 0.8, (N, 1))
    o = c + unit(rng.normal(size=(N, 3))) * r * rng.uniform(0.0, 4.0, (N, 1))
    tgt = c + unit(rng.normal(size=(N, 3))) * r * rng.uniform(0.0, 1.2, (N, 1))
    d = unit(tgt - o)
    tmax = np.where(rng.uniform(size=(N, 1)) < 0.3, rng.uniform(0.05, 3, (N, 1)), np.inf)
    man["sphere"] = table("sphere", np.hstack([c, r, o, d, np.full((N, 1), 1e-7), tmax]))
    # to_world (src/vector.h:314-326), incl. the n.z < -1+1e-6 branch
    n = unit(rng.normal(size=(N, 3)))
    n[::10] = np.array([0, 0, -1.0]); n[1::10] = unit(np.array([[1e-4, 0, -1.0]]))
    man["to_world"] = table("to_world", np.hstack([n, rng.normal(size=(N, 3))]))
    man["hemicos"] = table("hemicos", np.arange(256, dtype=np.float64) * 7 + 1)
    # materials (src/material.cpp:76-98, src/materials/*.inl): 12 tags x rows
    M = 192
    rows = []
    for tag in range(12):
        gn = unit(rng.normal(size=(M, 3)))
        sn = unit(gn + 0.3 * rng.normal(size=(M, 3)))
        sn[::6] = gn[::6]
        sn[1::12] = -sn[1::12]                       # shading normal on the other side
        din = unit(gn * rng.uniform(0.02, 1, (M, 1)) + 0.7 * rng.normal(size=(M, 3)))
        din[::8] = unit(-gn[::8] + 0.5 * rng.normal(size=(len(gn[::8]), 3)))     # below the surface
        dout = unit(gn * rng.uniform(-0.2, 1, (M, 1)) + 0.7 * rng.normal(size=(M, 3)))
        # near-specular dir_out for the glossy lobes
        nside = np.where(np.sum(din * sn, 1, keepdims=True) < 0, -sn, sn)
        refl = -din + 2 * np.sum(din * nside, 1, keepdims=True) * nside
        dout[::3] = unit(refl[::3] + 0.15 * rng.normal(size=(len(refl[::3]), 3)))
        col = rng.uniform(0.05, 0.95, (M, 3))
        if tag in (1, 2):
            p0 = rng.choice([1.0, 1.33, 1.5, 2.4], (M, 1))
        elif tag in (3, 4, 5):
            p0 = rng.choice([1.0, 5.0, 50.0, 500.0], (M, 1))
        else:
            p0 = rng.uniform(0, 1, (M, 1))
        p1 = rng.uniform(0, 1, (M, 1))
        uv = rng.uniform(-1.5, 2.5, (M, 2))
        recpdf = np.where(rng.uniform(size=(M, 1)) < 0.3, 1.0, rng.uniform(0.01, 3, (M, 1)))
        seed = rng.integers(1, 2**31 - 1, (M, 1)).astype(np.float64)
        texk = (rng.uniform(size=(M, 1)) < 0.25).astype(np.float64)
        tx = np.hstack([rng.choice([1.0, 2.0, 0.5], (M, 2)), rng.choice([0.0, 0.25, -0.4], (M, 2))])
        rows.append(np.hstack([np.full((M, 1), float(tag)), col, p0, p1, gn, sn, uv, din, dout, recpdf, seed, texk, tx]))
    man["material"] = table("material", np.vstack(rows))
    # texture bilinear (src/texture.cpp:3-25), incl. the wrap seam and negative uv
    T = 512
    uv = rng.uniform(-2, 3, (T, 2))
    uv[::16, 0] = 0.999; uv[1::16, 1] = 0.9999; uv[2::16] = 0.0; uv[3::16] = 1.0
    tx = np.hstack([rng.choice([1.0, 2.0, 0.5], (T, 2)), rng.choice([0.0, 0.25, -0.4], (T, 2))])
    man["texture"] = table("texture", np.hstack([uv, tx]))
    # light sampling / pdf (src/shape.cpp:125-169, src/light.cpp:32-56)
    L = 512
    kind = (np.arange(L) % 2).astype(np.float64).reshape(L, 1)
    geo = rng.uniform(-1, 1, (L, 9))
    geo[:, 3] = np.where(kind[:, 0] == 0, rng.uniform(0.05, 0.5, L), geo[:, 3])   # radius for spheres
    nn = unit(rng.normal(size=(L, 3, 3))).reshape(L, 9)
    ref = rng.uniform(-2, 2, (L, 3))
    seed = rng.integers(1, 2**31 - 1, (L, 1)).astype(np.float64)
    qp = rng.uniform(-1, 1, (L, 3)); qn = unit(rng.normal(size=(L, 3)))
    man["light"] = table("light", np.hstack([kind, geo, nn, ref, seed, np.ones((L, 1)), qp, qn]))
    # construct_bvh (src/bvh.cpp:8-45) on 1, 2, 3, 17 and 200 boxes
    for nb in (1, 2, 3, 17, 200):
        c = rng.uniform(-1, 1, (nb, 3)); e = rng.uniform(0.01, 0.2, (nb, 3))
        man[f"bvh-{nb}"] = table(f"bvh-{nb}", np.hstack([c - e, c + e]))


def make_scene_goldens(names, man):
    os.makedirs(RENDER, exist_ok=True)
    rng = np.random.default_rng(99)
    for name in names:
        xml = os.path.join(SCENES, name + ".xml")
        run("flatten", xml, os.path.join(SCENES, name + ".tkscene"))
        depths = [0, 1, 5, 50] if name == "cbox" else [50] if name != "meshlight" else [5]
        for d in depths:
            out = os.path.join(RENDER, f"{name}_d{d}.f64")
            run("render", xml, d, 4, out)
            a = np.fromfile(out, "<f8")
            man[f"render/{name}_d{d}"] = {"w": int(a[0]), "h": int(a[1]), "mean": float(a[2:].mean())}
        # scene_intersect / scene_occluded on random rays (src/scene.cpp:25-64)
        R = 2048
        o = rng.uniform(-0.95, 0.95, (R, 3))
        o[::4] = np.array([0, 0, 3.9])
        d = unit(rng.normal(size=(R, 3)))
        d[::4] = unit(rng.uniform(-0.3, 0.3, (len(d[::4]), 3)) + np.array([0, 0, -1.0]))
        tmax = np.where(rng.uniform(size=(R, 1)) < 0.3, rng.uniform(0.05, 2, (R, 1)), np.inf)
        man[f"isect-{name}"] = table(f"isect-{name}", np.hstack([o, d, np.full((R, 1), 1e-7), tmax]), xml)
        # path_tracing() on single camera-like rays with private seeds (src/integrator/path_tracing.h:5-111)
        P = 512
        o = np.tile(np.array([[0, 0, 3.9]]), (P, 1))
        d = unit(rng.uniform(-0.33, 0.33, (P, 3)) * np.array([1, 1, 0]) + np.array([0, 0, -1.0]))
        seed = rng.integers(1, 2**31 - 1, (P, 1)).astype(np.float64)
        for dd in ([5, 50] if name in ("cbox", "mats") else [50]):
            man[f"pt-{name}-d{dd}"] = table(f"pt-{name}-d{dd}", np.hstack([o, d, seed]), xml, dd)


INTEGRATOR_TABLES = {1: "ptraw", 2: "ptone", 3: "ptpow"}  # TakeRenderOpts.integrator -> table prefix


def make_integrator_tables():
    """The three integrators the reference defines but never calls (src/integrator/path_tracing.h:114 raw, :161
    one-sample MIS, :274 one-sample MIS with power-based light picking) on the rays and seeds of the committed pt-*
    tables: same inputs (pt-<scene>-d<depth>_in.f64), outputs <prefix>-<scene>-d<depth>_out.f64 = radiance3 +
    the next random_real (draw count).  `python oracle/gen_golden.py integrators` adds them to the manifest without
    touching the other fixtures."""
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    for key in sorted(k for k in man if k.startswith("pt-")):
        _, name, d = key.split("-")
        depth = int(d[1:])
        xml = os.path.join(SCENES, name + ".xml")
        fin = os.path.join(TABLES, f"{key}_in.f64")
        for mode, prefix in INTEGRATOR_TABLES.items():
            fout = os.path.join(TABLES, f"{prefix}-{name}-{d}_out.f64")
            run("pt", xml, depth, fin, fout, mode)
            out = np.fromfile(fout, "<f8").reshape(-1, 4)
            man[f"{prefix}-{name}-{d}"] = {"in": f"{key}_in.f64", "out_len": int(out.size),
                                           "mean_radiance": float(out[:, :3].mean())}
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


def make_egress():
    """tests/golden/egress/: float images and the EXR files the reference's imwrite (tinyexr, half, ZIP) makes of them"""
    d = os.path.join(GOLD, "egress")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(99)
    for name, (w, h) in (("zip_40x37", (40, 37)), ("none_12x9", (12, 9))):
        img = rng.uniform(0.0, 4.0, (h, w, 3)) ** 3
        img[0, 0] = (0.0, 1e-9, 65504.0)      # zero, a half denormal after rounding, the largest half
        img[0, 1] = (1e5, 6.1e-5, 1.00048828125)  # overflow -> inf, smallest normal region, a rounding tie (1 + 2^-11)
        img[1, 0] = (0.1, 1.0 / 3.0, 2049.0)   # 2049 is a tie between 2048 and 2050
        np.concatenate([[w, h], img.reshape(-1)]).astype("<f8").tofile(os.path.join(d, name + ".f64"))
        run("imwrite", os.path.join(d, name + ".f64"), os.path.join(d, name + ".exr"))


PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "<i2", "ushort": "<u2", "int": "<i4", "uint": "<u4", "float": "<f4",
             "double": "<f8", "int8": "i1", "uint8": "u1", "int16": "<i2", "uint16": "<u2", "int32": "<i4",
             "uint32": "<u4", "float32": "<f4", "float64": "<f8"}


def write_ply_general(path, vertex_props, vertex_cols, face_pre, count_type, index_type, faces, face_post=(),
                      face_first=False, extra_element=False, comments=()):
    """binary little-endian PLY with any scalar types / extra properties / element order (tests of the device decode).
    vertex_props: [(type name, property name)], vertex_cols: {property name: column}; face_pre / face_post: scalar
    properties around the `vertex_indices` list, [(type name, property name, column)]."""
    nv = len(next(iter(vertex_cols.values())))
    vdt = np.dtype([(n, PLY_TYPES[t]) for t, n in vertex_props])
    vert = np.zeros(nv, vdt)
    for _, n in vertex_props:
        vert[n] = vertex_cols[n]
    fdt = np.dtype([(n, PLY_TYPES[t]) for t, n, _ in face_pre] + [("cnt", PLY_TYPES[count_type]), ("idx", PLY_TYPES[index_type], 3)] +
                   [(n, PLY_TYPES[t]) for t, n, _ in face_post])
    face = np.zeros(len(faces), fdt)
    face["cnt"] = 3
    face["idx"] = faces
    for _, n, col in list(face_pre) + list(face_post):
        face[n] = col
    vhdr = [f"element vertex {nv}"] + [f"property {t} {n}" for t, n in vertex_props]
    fhdr = ([f"element face {len(faces)}"] + [f"property {t} {n}" for t, n, _ in face_pre] +
            [f"property list {count_type} {index_type} vertex_indices"] + [f"property {t} {n}" for t, n, _ in face_post])
    ehdr = ["element edge 3", "property int vertex1", "property int vertex2"] if extra_element else []
    body = [face.tobytes(), vert.tobytes()] if face_first else [vert.tobytes(), face.tobytes()]
    hdr = ["ply", "format binary_little_endian 1.0"] + [f"comment {c}" for c in comments]
    hdr += (fhdr + vhdr) if face_first else (vhdr + fhdr)
    hdr += ehdr + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        for b in body:
            f.write(b)
        if extra_element:
            f.write(np.arange(6, dtype="<i4").tobytes())


def ply_cases():
    """name -> (writer kwargs, to_world 4x4): the encodings src/parse/parse_ply.cpp:9-123 accepts"""
    rng = np.random.default_rng(2024)

    def geometry(nv, nf):
        pos = rng.uniform(-2.0, 2.0, (nv, 3))
        nrm = rng.normal(size=(nv, 3))
        uv = rng.uniform(0.0, 1.0, (nv, 2))
        faces = rng.integers(0, nv, (nf, 3))
        return pos, nrm, uv, faces

    def rot(ax, a):
        c, s_ = np.cos(a), np.sin(a)
        m = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, -s_, s_, c
        return m

    cases = {}
    eye = np.eye(4)
    affine = rot(0, 0.3) @ rot(1, -1.1) @ np.diag([1.5, 0.7, 2.0, 1.0])
    affine[:3, 3] = (0.25, -3.0, 1.75)
    persp = affine.copy()
    persp[3] = (0.01, -0.02, 0.03, 1.25)  # w != 1: xform_point divides by it
    # 1: the writer of scenes/*.ply: float positions, uchar count, int indices
    pos, nrm, uv, faces = geometry(300, 500)
    cases["f32_plain"] = (dict(vertex_props=[("float", "x"), ("float", "y"), ("float", "z")],
                               vertex_cols={"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]}, face_pre=[], count_type="uchar",
                               index_type="int", faces=faces), eye)
    # 2: normals + uvs interleaved with properties nobody asked for, an affine to_world
    pos, nrm, uv, faces = geometry(257, 401)
    cols = {"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "nx": nrm[:, 0], "ny": nrm[:, 1], "nz": nrm[:, 2], "u": uv[:, 0],
            "v": uv[:, 1], "red": rng.integers(0, 255, 257), "quality": rng.uniform(size=257)}
    cases["f32_normals_uvs_affine"] = (dict(
        vertex_props=[("float", "x"), ("float", "y"), ("float", "z"), ("uchar", "red"), ("float", "nx"), ("float", "ny"),
                      ("float", "nz"), ("double", "quality"), ("float", "u"), ("float", "v")],
        vertex_cols=cols, face_pre=[], count_type="uchar", index_type="uint", faces=faces, comments=("made by gen_golden",)), affine)
    # 3: everything double, homogeneous to_world, a zero normal (normalize() returns the zero vector)
    pos, nrm, uv, faces = geometry(129, 200)
    nrm[7] = 0.0
    cols = {"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "nx": nrm[:, 0], "ny": nrm[:, 1], "nz": nrm[:, 2], "u": uv[:, 0], "v": uv[:, 1]}
    cases["f64_all_projective"] = (dict(
        vertex_props=[("double", n) for n in ("x", "y", "z", "nx", "ny", "nz", "u", "v")], vertex_cols=cols, face_pre=[],
        count_type="uint8", index_type="int32", faces=faces), persp)
    # 4: 16-bit indices, scalar face properties on both sides of the list, faces ahead of the vertices, a trailing element
    pos, nrm, uv, faces = geometry(1000, 777)
    cases["u16_faces_first"] = (dict(
        vertex_props=[("float32", "x"), ("float32", "y"), ("float32", "z"), ("float32", "u"), ("float32", "v")],
        vertex_cols={"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "u": uv[:, 0], "v": uv[:, 1]},
        face_pre=[("uchar", "flags", rng.integers(0, 255, 777))], count_type="uchar", index_type="ushort", faces=faces,
        face_post=[("float", "area", rng.uniform(size=777))], face_first=True, extra_element=True), affine)
    # 5: 8-bit indices, a 16-bit count
    pos, nrm, uv, faces = geometry(100, 64)
    cases["i8_indices"] = (dict(vertex_props=[("float", "x"), ("float", "y"), ("float", "z"), ("float", "nx"), ("float", "ny"), ("float", "nz")],
                                vertex_cols={"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2], "nx": nrm[:, 0], "ny": nrm[:, 1], "nz": nrm[:, 2]},
                                face_pre=[], count_type="ushort", index_type="char", faces=faces), rot(2, 0.5))
    pos, nrm, uv, faces = geometry(200, 90)
    cases["u8_i16"] = (dict(vertex_props=[("double", "x"), ("double", "y"), ("double", "z")],
                            vertex_cols={"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]}, face_pre=[], count_type="int",
                            index_type="short", faces=faces), eye)
    return cases


def make_ply():
    """tests/golden/ply/: PLY files in every encoding the reference's parse_ply reads, its to_world matrices, and the
    TriangleMesh arrays the reference's own parser (ref_harness ply) made of them"""
    d = os.path.join(GOLD, "ply")
    os.makedirs(d, exist_ok=True)
    for name, (kw, xf) in ply_cases().items():
        write_ply_general(os.path.join(d, name + ".ply"), **kw)
        np.asarray(xf, "<f8").reshape(-1).tofile(os.path.join(d, name + "_xform.f64"))
        run("ply", os.path.join(d, name + ".ply"), os.path.join(d, name + "_xform.f64"), os.path.join(d, name + "_mesh.f64"))
        out = np.fromfile(os.path.join(d, name + "_mesh.f64"), "<f8")
        print(f"ply/{name}: {int(out[0])} vertices, {int(out[1])} faces, normals {int(out[2])}, uvs {int(out[3])}")


def serialized_blob(version, flags, name, pos, nrm, uv, col, faces):
    """one sub-mesh of Mitsuba's serialized format: [u16 magic][u16 version][zlib: flags, (v4: name NUL), nv, nf, blocks]"""
    import zlib

    t = "<f8" if flags & 0x2000 else "<f4"
    body = struct.pack("<I", flags)
    if version == 4:
        body += name.encode() + b"\0"
    body += struct.pack("<QQ", len(pos), len(faces)) + np.asarray(pos, t).tobytes()
    if flags & 0x0001:
        body += np.asarray(nrm, t).tobytes()
    if flags & 0x0002:
        body += np.asarray(uv, t).tobytes()
    if flags & 0x0008:
        body += np.asarray(col, t).tobytes()
    body += np.asarray(faces, "<i4").tobytes()
    return struct.pack("<HH", 0x041C, version) + zlib.compress(body, 6)


def write_serialized(path, version, blobs):
    """the sub-meshes back to back, then their offsets (u64 in version 4, u32 in version 3) and the u32 count"""
    offs, data = [], b""
    for b in blobs:
        offs.append(len(data))
        data += b
    data += b"".join(struct.pack("<Q" if version == 4 else "<I", o) for o in offs) + struct.pack("<I", len(blobs))
    with open(path, "wb") as f:
        f.write(data)


def serialized_cases():
    """name -> (version, [blob kwargs], [(shape_index, to_world)])"""
    rng = np.random.default_rng(777)

    def geo(nv, nf):
        return dict(pos=rng.uniform(-2, 2, (nv, 3)), nrm=rng.normal(size=(nv, 3)), uv=rng.uniform(0, 1, (nv, 2)),
                    col=rng.uniform(0, 1, (nv, 3)), faces=rng.integers(0, nv, (nf, 3)))

    eye = np.eye(4)
    aff = np.array([[0.0, -1.5, 0.0, 0.5], [2.0, 0.0, 0.0, -1.0], [0.0, 0.0, 0.75, 3.0], [0.0, 0.0, 0.0, 1.0]])
    aff[:3, :3] = aff[:3, :3] @ np.array([[np.cos(0.4), 0, np.sin(0.4)], [0, 1, 0], [-np.sin(0.4), 0, np.cos(0.4)]])
    proj = aff.copy()
    proj[3] = (0.02, 0.01, -0.03, 0.9)
    g = geo(150, 220)
    g["nrm"][5] = 0.0  # normalize() returns the zero vector
    return {
        "v4_f32_three_meshes": (4, [dict(flags=0x1000, name="plain", **geo(120, 200)),
                                    dict(flags=0x1000 | 0x0001 | 0x0002, name="normals and uvs", **geo(257, 300)),
                                    dict(flags=0x1000 | 0x0001 | 0x0002 | 0x0008 | 0x0010, name="", **geo(64, 99))],
                                [(0, eye), (1, aff), (2, proj)]),
        "v3_f64_two_meshes": (3, [dict(flags=0x2000 | 0x0002 | 0x0008, name="ignored", **geo(90, 131)),
                                  dict(flags=0x2000 | 0x0001, name="ignored", **g)], [(0, aff), (1, proj)]),
    }


def make_serialized():
    """tests/golden/serialized/: Mitsuba-serialized files (versions 3 and 4, float and double, every block combination,
    several sub-meshes) and the TriangleMesh arrays the reference's own parse_serialized made of each sub-mesh"""
    d = os.path.join(GOLD, "serialized")
    os.makedirs(d, exist_ok=True)
    for name, (version, blobs, picks) in serialized_cases().items():
        path = os.path.join(d, name + ".serialized")
        write_serialized(path, version, [serialized_blob(version, **b) for b in blobs])
        for idx, xf in picks:
            key = f"{name}_{idx}"
            np.asarray(xf, "<f8").reshape(-1).tofile(os.path.join(d, key + "_xform.f64"))
            run("serialized", path, idx, os.path.join(d, key + "_xform.f64"), os.path.join(d, key + "_mesh.f64"))
            out = np.fromfile(os.path.join(d, key + "_mesh.f64"), "<f8")
            print(f"serialized/{key}: {int(out[0])} vertices, {int(out[1])} faces, normals {int(out[2])}, uvs {int(out[3])}")


def main():
    if not os.path.exists(HARNESS):
        raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` (needs /root/reference)")
    if sys.argv[1:] == ["egress"]:
        make_egress()
        return
    if sys.argv[1:] == ["integrators"]:
        make_integrator_tables()
        return
    if sys.argv[1:] == ["disney"]:
        make_disney()
        return
    if sys.argv[1:] == ["ply"]:
        make_ply()
        return
    if sys.argv[1:] == ["serialized"]:
        make_serialized()
        return
    man = {}
    names = make_scene_files()
    make_tables(man)
    make_scene_goldens(names, man)
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(GOLD) for fn in fns)
    print(f"golden fixtures written: {tot / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
