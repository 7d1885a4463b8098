"""Procedural benchmark scenes: random triangle soups inside a 5-wall box with one quad light.

Everything is generated from numpy's PCG64 (`default_rng(seed)`), so the GPU box and the authoring container
build identical inputs.  The scenes are expressed in the reference's own scene model (take_amd.scene.SceneData
mirrors src/scene.h): rectangles are the parser's 2x2 quads with vertex normals and uvs
(src/parse/parse_scene.cpp:891-927), the soup is a mesh with unshared vertices and face normals
(`faceNormals=true`: no vertex-normal array), the light is an emissive rectangle = two DiffuseAreaLights.
"""
import dataclasses
import math

import numpy as np

from . import cdefs as D
from .scene import SceneData


def vfov_from_xfov(fov_x_deg, width, height):
    """fovAxis = x -> vertical fov, as parse_sensor does (src/parse/parse_scene.cpp:366-370)."""
    return math.degrees(2 * math.atan(math.tan(math.radians(fov_x_deg) / 2) * height / float(width)))


def _quad(center, ux, uy, n):
    """2x2 parser rectangle transformed: corners centre -ux-uy, +ux-uy, +ux+uy, -ux+uy; normal n"""
    c, ux, uy = map(lambda v: np.asarray(v, np.float64), (center, ux, uy))
    pos = np.stack([c - ux - uy, c + ux - uy, c + ux + uy, c - ux + uy])
    idx = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64)
    nrm = np.tile(np.asarray(n, np.float64), (4, 1))
    return pos, idx, nrm, uv


def box_with_light(sd: SceneData, white, red, green, light_half=0.3, radiance=(17.0, 12.0, 4.0)):
    """back, floor, ceiling, left (red), right (green) walls of the [-1,1]^3 box + a quad light under the ceiling"""
    walls = [
        ((0, 0, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1), white),   # back
        ((0, -1, 0), (1, 0, 0), (0, 0, -1), (0, 1, 0), white),  # floor
        ((0, 1, 0), (1, 0, 0), (0, 0, 1), (0, -1, 0), white),   # ceiling
        ((-1, 0, 0), (0, 0, -1), (0, 1, 0), (1, 0, 0), red),    # left
        ((1, 0, 0), (0, 0, 1), (0, 1, 0), (-1, 0, 0), green),   # right
    ]
    for c, ux, uy, n, mat in walls:
        pos, idx, nrm, uv = _quad(c, ux, uy, n)
        sd.add_mesh(pos, idx, mat, normals=nrm, uvs=uv)
    pos, idx, nrm, uv = _quad((0, 0.99, 0), (light_half, 0, 0), (0, 0, light_half), (0, -1, 0))
    sd.add_mesh(pos, idx, white, normals=nrm, uvs=uv, emission=radiance)


def soup_triangles(n, seed=1234, half=0.9, jitter=0.02):
    """n triangles: centres ~U(-half,half)^3, vertex offsets ~U(-jitter,jitter)^3; unshared vertices"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-half, half, (n, 1, 3))
    v = c + rng.uniform(-jitter, jitter, (n, 3, 3))
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.int32).reshape(n, 3)


def sky_envmap(width=2048, height=1024, sun_dir=(0.35, 0.45, 0.82), sun_radiance=400.0, sun_halfangle_deg=3.0):
    """Procedural equirectangular environment map (configs[2]: "procedural sky + sun"): y up, row 0 = zenith,
    u = atan2(z, x) / 2pi + 1/2.  Blue-to-white sky gradient above the horizon, dim ground below, one small sun."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    theta = v[:, None] * np.pi
    phi = (u[None, :] - 0.5) * 2.0 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], -1)
    up = np.clip(d[..., 1], 0.0, 1.0)
    sky = (1.0 - up)[..., None] * np.array([1.0, 1.0, 1.0]) + up[..., None] * np.array([0.3, 0.5, 1.0])
    img = np.where(d[..., 1:2] >= 0.0, 0.8 * sky, np.array([0.08, 0.07, 0.06]))
    s = np.asarray(sun_dir, np.float64)
    s = s / np.linalg.norm(s)
    img = img + (d @ s >= np.cos(np.radians(sun_halfangle_deg)))[..., None] * sun_radiance * np.array([1.0, 0.9, 0.7])
    return np.ascontiguousarray(img, np.float64)


def soup_scene(n_tris, width, height, spp, seed=1234, jitter=None, max_depth=50, materials="diffuse", envmap=None):
    """BASELINE configs[1] (100k) / [2] (1M): soup + box + quad light, camera (0,0,3.9), fov 39 deg on the x axis,
    background 0.  envmap = (w, h): adds the procedural sky of sky_envmap as an importance-sampled environment light
    (configs[2]'s "env-map IBL" — an extension, the reference has no such light); it shines in through the open
    front of the box."""
    if jitter is None:
        jitter = 0.02 if n_tris <= 200_000 else 0.008
    sd = SceneData(width=width, height=height, lookfrom=(0.0, 0.0, 3.9), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                   vfov=vfov_from_xfov(39.0, width, height), background=(0.0, 0.0, 0.0), spp=spp, max_depth=max_depth)
    white = sd.add_material(D.MAT_DIFFUSE, (0.73, 0.73, 0.73))
    red = sd.add_material(D.MAT_DIFFUSE, (0.65, 0.05, 0.05))
    green = sd.add_material(D.MAT_DIFFUSE, (0.12, 0.45, 0.15))
    box_with_light(sd, white, red, green)
    pos, idx = soup_triangles(n_tris, seed, 0.9, jitter)
    if materials == "diffuse":
        sd.add_mesh(pos, idx, white)
    else:
        # divergence stress (configs[4]): round-robin over the reference's real BSDFs
        mats = [white,
                sd.add_material(D.MAT_PLASTIC, (0.2, 0.3, 0.8), (1.5,)),
                sd.add_material(D.MAT_PHONG, (0.5, 0.6, 0.3), (20.0,)),
                sd.add_material(D.MAT_BLINN_PHONG, (0.3, 0.4, 0.7), (40.0,)),
                sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (5.0,)),
                sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (50.0,)),
                sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (500.0,))]
        k = len(mats)
        tri = pos.reshape(-1, 3, 3)
        for j, m in enumerate(mats):
            part = tri[j::k]
            sd.add_mesh(part.reshape(-1, 3), np.arange(3 * len(part), dtype=np.int32).reshape(-1, 3), m)
    if envmap is not None:
        sd.add_envmap(sky_envmap(int(envmap[0]), int(envmap[1])))
    return sd


def instanced_scene(n_instances, tris_per_mesh, width, height, spp, seed=4321, max_depth=50, flatten=False):
    """BASELINE configs[4]: `n_instances` placements of one `tris_per_mesh`-triangle mesh under random rigid
    transforms, materials round-robin over the reference's real BSDFs (Diffuse, Plastic, Phong, BlinnPhong, three
    BlinnPhongMicrofacet exponents) — the divergence stress.  The reference has no instancing.
    Default: true instancing (TakeInstance, an extension): ONE prototype mesh in object space + one transform and
    material per placement, traversed on two levels.  flatten=True: the same geometry expanded to world-space
    triangles (SceneData.flattened) — what the instanced render is specified to equal, to fp rounding, and the only
    form the reference's own scene model can express.  Same box, light and camera as soup_scene."""
    sd = SceneData(width=width, height=height, lookfrom=(0.0, 0.0, 3.9), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                   vfov=vfov_from_xfov(39.0, width, height), background=(0.0, 0.0, 0.0), spp=spp, max_depth=max_depth)
    white = sd.add_material(D.MAT_DIFFUSE, (0.73, 0.73, 0.73))
    red = sd.add_material(D.MAT_DIFFUSE, (0.65, 0.05, 0.05))
    green = sd.add_material(D.MAT_DIFFUSE, (0.12, 0.45, 0.15))
    box_with_light(sd, white, red, green)
    mats = [white,
            sd.add_material(D.MAT_PLASTIC, (0.2, 0.3, 0.8), (1.5,)),
            sd.add_material(D.MAT_PHONG, (0.5, 0.6, 0.3), (20.0,)),
            sd.add_material(D.MAT_BLINN_PHONG, (0.3, 0.4, 0.7), (40.0,)),
            sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (5.0,)),
            sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (50.0,)),
            sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (500.0,))]
    rng = np.random.default_rng(seed)
    # the prototype: a small cloud of triangles around the origin (radius ~0.08)
    proto, pidx = soup_triangles(tris_per_mesh, seed + 1, 0.07, 0.012)
    # random rigid transforms: rotation from a unit quaternion, translation inside the box
    q = rng.normal(size=(n_instances, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    rot = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                    np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                    np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)
    trans = rng.uniform(-0.85, 0.85, (n_instances, 3))
    mesh = sd.add_prototype(proto, pidx, white)
    for i in range(n_instances):
        sd.add_instance(mesh, np.concatenate([rot[i], trans[i][:, None]], axis=1), mats[i % len(mats)])
    return sd.flattened() if flatten else sd


BURLEY_BSDF_ORDER = ("specular_transmission", "metallic", "subsurface", "specular", "roughness", "specular_tint",
                     "anisotropic", "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss", "eta")


def principled(**kw):
    """TakeMaterial.param of a DisneyBSDF / BURLEY_BSDF material (defaults: the reference parser's,
    src/parse/parse_scene.cpp:635-700)"""
    d = dict(specular_transmission=0.0, metallic=0.0, subsurface=0.0, specular=0.5, roughness=0.5, specular_tint=0.0,
             anisotropic=0.0, sheen=0.0, sheen_tint=0.5, clearcoat=0.0, clearcoat_gloss=1.0, eta=1.5)
    d.update(kw)
    return tuple(float(d[k]) for k in BURLEY_BSDF_ORDER)


def _cube(center, half):
    """closed box with outward-wound faces, no vertex normals (so the geometric orientation decides front / back)"""
    c = np.asarray(center, np.float64)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) * half + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = []
    for a, b, cc, d in quads:
        idx += [[a, b, cc], [a, cc, d]]
    return v, np.array(idx, np.int32)


def burley_scene(width=96, height=96, spp=8, real=True, max_depth=8):
    """A Cornell-style box holding one object per Disney material — metal (anisotropic) sphere, glass sphere, glass
    cube (triangle back faces), sheen sphere, two principled spheres (one transmissive with clearcoat, one metallic
    and tinted) and a clearcoat-only sphere.  real=True uses tags 12..16 (the Burley lobes); real=False uses the
    reference's tags 7..11 with the same parameters (Lambert clones upstream; TakeBuildOpts.burley_lobes maps them
    to the former)."""
    from . import cdefs as D

    sd = SceneData(width=width, height=height, lookfrom=(0.0, 0.0, 3.4), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0),
                   vfov=38.0, background=(0.0, 0.0, 0.0), spp=spp, max_depth=max_depth)
    white = sd.add_material(D.MAT_DIFFUSE, (0.73, 0.73, 0.73))
    red = sd.add_material(D.MAT_DIFFUSE, (0.65, 0.05, 0.05))
    green = sd.add_material(D.MAT_DIFFUSE, (0.12, 0.45, 0.15))
    box_with_light(sd, white, red, green, light_half=0.35)
    off = 5 if real else 0
    metal = sd.add_material(D.MAT_DISNEY_METAL + off, (0.95, 0.64, 0.54), (0.35, 0.7))
    glass = sd.add_material(D.MAT_DISNEY_GLASS + off, (0.95, 0.97, 1.0), (0.15, 0.0, 1.5))
    glass2 = sd.add_material(D.MAT_DISNEY_GLASS + off, (1.0, 0.85, 0.7), (0.3, 0.4, 1.33))
    coat = sd.add_material(D.MAT_DISNEY_CLEARCOAT + off, (0.0, 0.0, 0.0), (0.6,))
    sheen = sd.add_material(D.MAT_DISNEY_SHEEN + off, (0.3, 0.5, 0.9), (0.7,))
    pr1 = sd.add_material(D.MAT_DISNEY_BSDF + off, (0.8, 0.3, 0.2),
                          principled(specular_transmission=0.6, roughness=0.25, clearcoat=0.8, clearcoat_gloss=0.5,
                                     sheen=0.4, eta=1.45))
    pr2 = sd.add_material(D.MAT_DISNEY_BSDF + off, (0.9, 0.75, 0.3),
                          principled(metallic=0.8, roughness=0.4, anisotropic=0.6, specular_tint=0.5, subsurface=0.3))
    sd.add_sphere((-0.6, -0.69, -0.3), 0.3, metal)
    sd.add_sphere((0.1, -0.67, 0.35), 0.32, glass)
    sd.add_sphere((0.65, -0.74, -0.35), 0.25, sheen)
    sd.add_sphere((-0.55, 0.0, -0.55), 0.27, pr1)
    sd.add_sphere((0.55, 0.05, -0.5), 0.27, pr2)
    sd.add_sphere((0.0, 0.45, -0.6), 0.2, coat)
    pos, idx = _cube((-0.15, -0.79, -0.5), 0.2)  # nothing coincides with the floor (exact ties: DESIGN.md §6)
    sd.add_mesh(pos, idx, glass2)
    return sd


def with_burley_lobes(sd: SceneData):
    """copy of `sd` whose Disney materials (tags 7..11) carry the tags of the real lobes (12..16) — what
    TakeBuildOpts.burley_lobes does inside take_hip_scene_create"""
    import copy
    from . import cdefs as D

    out = copy.copy(sd)
    out.materials = [dataclasses.replace(m, tag=m.tag + 5) if D.MAT_DISNEY_METAL <= m.tag <= D.MAT_DISNEY_BSDF else m
                     for m in sd.materials]
    return out


def write_reference_inputs(sd: SceneData, directory, name="scene"):
    """Write `sd` (a soup_scene) as XML + binary PLY in the reference's dialect, for oracle/_ref/ref_harness
    (bench.py's `cpu_baseline` leg with kind = "reference").  Only the features soup_scene uses."""
    import os
    import struct

    os.makedirs(directory, exist_ok=True)
    lines = ['<scene version="0.5.0">',
             f'<sensor type="perspective"><float name="fov" value="{sd.vfov!r}"/><string name="fovAxis" value="y"/>',
             f'<transform name="toWorld"><lookat origin="{sd.lookfrom[0]!r},{sd.lookfrom[1]!r},{sd.lookfrom[2]!r}" '
             f'target="{sd.lookat[0]!r},{sd.lookat[1]!r},{sd.lookat[2]!r}" up="{sd.up[0]!r},{sd.up[1]!r},{sd.up[2]!r}"/></transform>',
             f'<sampler type="independent"><integer name="sampleCount" value="{sd.spp}"/></sampler>',
             f'<film type="hdrfilm"><integer name="width" value="{sd.width}"/><integer name="height" value="{sd.height}"/></film></sensor>',
             f'<background><rgb name="radiance" value="{sd.background[0]!r} {sd.background[1]!r} {sd.background[2]!r}"/></background>']
    tagname = {D.MAT_DIFFUSE: "diffuse", D.MAT_PLASTIC: "plastic", D.MAT_PHONG: "phong", D.MAT_BLINN_PHONG: "blinn",
               D.MAT_BLINN_PHONG_MICROFACET: "blinn_microfacet", D.MAT_MIRROR: "mirror"}
    for i, m in enumerate(sd.materials):
        extra = ""
        if m.tag in (D.MAT_PLASTIC,):
            extra = f'<float name="ior" value="{m.param[0]!r}"/>'
        elif m.tag in (D.MAT_PHONG, D.MAT_BLINN_PHONG, D.MAT_BLINN_PHONG_MICROFACET):
            extra = f'<float name="exponent" value="{m.param[0]!r}"/>'
        lines.append(f'<bsdf type="{tagname[m.tag]}" id="m{i}"><rgb name="reflectance" '
                     f'value="{m.color[0]!r} {m.color[1]!r} {m.color[2]!r}"/>{extra}</bsdf>')
    light_of_shape = {l.shape_id: l for l in sd.lights if l.kind == 1}
    shape_base = 0
    for mi, mesh in enumerate(sd.meshes):
        ply = f"{name}_mesh{mi}.ply"
        nv, nf = mesh.positions.shape[0], mesh.indices.shape[0]
        props = ["property float x", "property float y", "property float z"]
        cols = [mesh.positions.astype("<f4")]
        if mesh.normals is not None:
            props += ["property float nx", "property float ny", "property float nz"]
            cols.append(mesh.normals.astype("<f4"))
        if mesh.uvs is not None:
            props += ["property float u", "property float v"]
            cols.append(mesh.uvs.astype("<f4"))
        hdr = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {nv}", *props,
                         f"element face {nf}", "property list uchar int vertex_indices", "end_header"]) + "\n"
        with open(os.path.join(directory, ply), "wb") as f:
            f.write(hdr.encode())
            f.write(np.concatenate(cols, axis=1).astype("<f4").tobytes())
            rec = np.zeros(nf, dtype=[("n", "u1"), ("i", "<i4", 3)])
            rec["n"] = 3
            rec["i"] = mesh.indices
            f.write(rec.tobytes())
        face_normals = "true" if mesh.normals is None else "false"
        em = ""
        if shape_base in light_of_shape:
            L = light_of_shape[shape_base].intensity
            em = f'<emitter type="area"><rgb name="radiance" value="{L[0]!r} {L[1]!r} {L[2]!r}"/></emitter>'
        lines.append(f'<shape type="ply"><string name="filename" value="{ply}"/><boolean name="faceNormals" '
                     f'value="{face_normals}"/><ref id="m{mesh.material_id}"/>{em}</shape>')
        shape_base += nf
    lines.append("</scene>")
    path = os.path.join(directory, name + ".xml")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path
