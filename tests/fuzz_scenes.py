"""Randomised scenes for the fuzz tests (tests/test_fuzz_scenes.py): every feature of the scene model at once, with the
degenerate inputs a parser can hand over — zero-area and needle triangles, duplicated triangles, a triangle seen
edge-on, nested and tiny spheres, all 17 material tags (the 12 of the reference + the 5 Burley lobes) with random
parameters, an image texture, emissive quads (with
vertex normals, as the reference requires), a sphere light, a point light, any scale from 1e-3 to 1e3."""
import numpy as np

from take_amd import cdefs as D
from take_amd import scenes
from take_amd.scene import Light, SceneData


def random_scene(seed, res=24):
    rng = np.random.default_rng(seed)
    scale = float(10.0 ** rng.uniform(-3, 3)) if seed % 3 == 0 else 1.0
    eye = np.array([0.0, 0.4, 3.2]) * scale
    sd = SceneData(width=res + int(rng.integers(0, 9)), height=res + int(rng.integers(0, 7)), lookfrom=tuple(eye),
                   lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=float(rng.uniform(30, 60)),
                   background=tuple(rng.uniform(0, 0.6, 3)), spp=2, max_depth=4)
    tex = rng.uniform(0.05, 0.95, (5, 7, 3))
    sd.images.append(tex)
    mats = []
    for tag in range(12):
        p = (float(rng.uniform(1.2, 1.8)),) if tag in (D.MAT_MIRROR, D.MAT_PLASTIC) else \
            (float(rng.uniform(2, 200)),) if tag in (D.MAT_PHONG, D.MAT_BLINN_PHONG, D.MAT_BLINN_PHONG_MICROFACET) else \
            (float(rng.uniform(0, 1)), float(rng.uniform(0, 1)))
        mats.append(sd.add_material(tag, tuple(rng.uniform(0.1, 0.9, 3)), p,
                                    tex_image=0 if rng.uniform() < 0.3 else None,
                                    uvxf=(float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(0, 1)),
                                          float(rng.uniform(0, 1)))))
    # the Burley lobes (tags 12..16) from a second stream, so that the geometry of a seed does not depend on them; about
    # a third of the material picks below are redirected to one of them
    rng2 = np.random.default_rng(seed + 100003)
    for tag in range(12, 17):
        p = rng2.uniform(0, 1, 12)
        if tag == D.MAT_BURLEY_GLASS:
            p[2] = rng2.uniform(1.1, 2.0)
        p[11] = rng2.uniform(1.1, 2.0)
        mats.append(sd.add_material(tag, tuple(rng2.uniform(0.1, 0.95, 3)), tuple(float(x) for x in p),
                                    tex_image=0 if rng2.uniform() < 0.3 else None,
                                    uvxf=(float(rng2.uniform(0.5, 3)), float(rng2.uniform(0.5, 3)), 0.25, 0.5)))
    reference_mats = list(mats[:12])
    mats = [m if rng2.uniform() > 0.35 else mats[12 + int(rng2.integers(0, 5))] for m in reference_mats]
    # a floor and a back wall so that paths bounce
    for c, ux, uy, n in (((0, -1, 0), (1.5, 0, 0), (0, 0, -1.5), (0, 1, 0)), ((0, 0, -1.2), (1.5, 0, 0), (0, 1.5, 0), (0, 0, 1))):
        pos, idx, nrm, uv = scenes._quad(c, ux, uy, n)
        sd.add_mesh(pos * scale, idx, mats[int(rng.integers(0, 12))], normals=nrm, uvs=uv)
    # random triangle clouds, one mesh per a few materials, with degenerate members
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(3, 60))
        c = rng.uniform(-0.8, 0.8, (n, 1, 3))
        v = c + rng.uniform(-0.3, 0.3, (n, 3, 3))
        v[0, 2] = v[0, 1]                                   # zero area: two equal vertices
        v[1, 2] = v[1, 0] + (v[1, 1] - v[1, 0]) * 0.5       # zero area: collinear
        if n > 4:
            v[2] = v[3]                                     # an exact duplicate
            v[4, 2] = v[4, 0] + (v[4, 1] - v[4, 0]) * (1 + 1e-9) + 1e-12  # a needle
        with_attr = rng.uniform() < 0.5
        nrm = uv = None
        if with_attr:
            nrm = rng.normal(size=(3 * n, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            uv = rng.uniform(-2, 3, (3 * n, 2))
            if n > 4:  # the exact duplicate carries the same attributes: which of the two wins a tie must not matter
                nrm[6:9], uv[6:9] = nrm[9:12], uv[9:12]  # (different attributes: tests/test_gpu_precision.py pins the rule)
        sd.add_mesh(v.reshape(-1, 3) * scale, np.arange(3 * n, dtype=np.int32).reshape(n, 3), mats[int(rng.integers(0, 12))],
                    normals=nrm, uvs=uv)
    for _ in range(int(rng.integers(0, 4))):
        c = rng.uniform(-0.7, 0.7, 3)
        r = float(rng.uniform(0.05, 0.4))
        sd.add_sphere(tuple(c * scale), r * scale, mats[int(rng.integers(0, 12))])
        if rng.uniform() < 0.3:  # a smaller sphere inside it
            sd.add_sphere(tuple(c * scale), 0.5 * r * scale, mats[int(rng.integers(0, 12))])
    # lights: an emissive quad (two triangle lights), sometimes a sphere light and a point light
    pos, idx, nrm, uv = scenes._quad((0, 1.3, 0.2), (0.4, 0, 0), (0, 0, 0.4), (0, -1, 0))
    sd.add_mesh(pos * scale, idx, mats[0], normals=nrm, uvs=uv, emission=tuple(rng.uniform(2, 12, 3)))
    if rng.uniform() < 0.5:
        sd.add_sphere((0.9 * scale, 0.6 * scale, 0.5 * scale), 0.12 * scale, mats[0], emission=tuple(rng.uniform(2, 12, 3)))
    if rng.uniform() < 0.4:
        sd.lights.append(Light(0, -1, (3.0, 3.0, 3.0), (0.0, 0.5 * scale, 0.0)))
    return sd, scale


def random_rays(sd, scale, n, seed, tmin):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-1.4, 1.4, (n, 3)) * scale
    d = rng.normal(size=(n, 3))
    k = n // 4
    o[:k] = np.asarray(sd.lookfrom)
    d[:k] = rng.uniform(-0.35, 0.35, (k, 3)) + np.array([0, -0.1, -1.0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.uniform(size=(n, 1)) < 0.3, rng.uniform(0.05, 3, (n, 1)) * scale, np.inf)
    return np.hstack([o, d, np.full((n, 1), tmin * scale), tmax])


def add_random_instances(sd, scale, seed):
    """1-3 prototype meshes (some with vertex normals and uvs, one zero-area face) under 2-6 placements each: affine
    transforms with scale, shear and — a third of them — a mirror (negative determinant); random material override"""
    rng = np.random.default_rng(1000 + seed)
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(2, 30))
        v = rng.uniform(-0.25, 0.25, (n, 3, 3))
        if n > 3:
            v[0, 2] = v[0, 1]
        nrm = uv = None
        if rng.uniform() < 0.5:
            nrm = rng.normal(size=(3 * n, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            uv = rng.uniform(0, 1, (3 * n, 2))
        proto = sd.add_prototype(v.reshape(-1, 3) * scale, np.arange(3 * n, dtype=np.int32).reshape(n, 3),
                                 int(rng.integers(0, 12)), normals=nrm, uvs=uv)
        for _ in range(int(rng.integers(2, 7))):
            L = np.eye(3) + rng.uniform(-0.6, 0.6, (3, 3))
            if rng.uniform() < 0.3:
                L[:, 0] *= -1
            if abs(np.linalg.det(L)) < 0.05:
                L = np.eye(3)
            t = rng.uniform(-0.7, 0.7, 3) * scale
            sd.add_instance(proto, np.concatenate([L, t[:, None]], axis=1), int(rng.integers(-1, 12)))
    return sd
