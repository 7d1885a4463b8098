"""Host-side scene model: the flattened form of TaKe's `Scene` (reference src/scene.h:13-33).

`SceneData` holds the same fields `TakeSceneDesc` (include/take_hip.h) points at, as numpy arrays,
and converts to the C struct with `to_desc()`.  `.tkscene` files (take_amd/host/take_sceneio.hpp)
are read and written here.
"""
import ctypes as C
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import cdefs as D


@dataclass
class Mesh:
    positions: np.ndarray  # (nv,3) f64
    indices: np.ndarray  # (nf,3) i32
    material_id: int
    normals: Optional[np.ndarray] = None  # (nv,3) f64
    uvs: Optional[np.ndarray] = None  # (nv,2) f64


@dataclass
class Sphere:
    center: tuple
    radius: float
    material_id: int


@dataclass
class Light:
    kind: int  # 0 point, 1 diffuse area
    shape_id: int
    intensity: tuple
    position: tuple = (0.0, 0.0, 0.0)


@dataclass
class Material:
    tag: int
    color: tuple = (0.5, 0.5, 0.5)
    tex_kind: int = 0
    tex_image: int = 0
    uvxf: tuple = (1.0, 1.0, 0.0, 0.0)  # uscale, vscale, uoffset, voffset
    param: tuple = (0.0,) * 12  # TakeMaterial::param


@dataclass
class SceneData:
    width: int = 256
    height: int = 256
    lookfrom: tuple = (0.0, 0.0, 0.0)
    lookat: tuple = (0.0, 0.0, -1.0)
    up: tuple = (0.0, 1.0, 0.0)
    vfov: float = 45.0
    background: tuple = (0.5, 0.5, 0.5)
    spp: int = 16
    max_depth: int = 50
    meshes: List[Mesh] = field(default_factory=list)
    spheres: List[Sphere] = field(default_factory=list)
    shape_kind: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    shape_ref: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    shape_face: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    shape_area_light: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    lights: List[Light] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    images: List[np.ndarray] = field(default_factory=list)  # (h,w,3) f64
    # EXTENSION (configs[4]): placements of prototype meshes, traversed on two levels — not flattened
    instance_mesh: List[int] = field(default_factory=list)
    instance_material: List[int] = field(default_factory=list)
    instance_xform: List[np.ndarray] = field(default_factory=list)  # (3,4) object -> world

    @property
    def n_shapes(self):
        return int(self.shape_kind.shape[0])

    # -- construction helpers mirroring parse_shape (reference src/parse/parse_scene.cpp:729-948) ----------
    def add_mesh(self, positions, indices, material_id, normals=None, uvs=None, emission=None):
        """Append a mesh; one Triangle shape per face, and — as the reference does
        (parse_scene.cpp:935-946) — one DiffuseAreaLight per face when `emission` is given."""
        mesh = Mesh(np.ascontiguousarray(positions, np.float64).reshape(-1, 3),
                    np.ascontiguousarray(indices, np.int32).reshape(-1, 3), int(material_id),
                    None if normals is None else np.ascontiguousarray(normals, np.float64).reshape(-1, 3),
                    None if uvs is None else np.ascontiguousarray(uvs, np.float64).reshape(-1, 2))
        mesh_id = len(self.meshes)
        self.meshes.append(mesh)
        nf = mesh.indices.shape[0]
        base = self.n_shapes
        al = np.full(nf, -1, np.int32)
        if emission is not None:
            first = len(self.lights)
            al = np.arange(first, first + nf, dtype=np.int32)
            for f in range(nf):
                self.lights.append(Light(1, base + f, tuple(emission)))
        self.shape_kind = np.concatenate([self.shape_kind, np.ones(nf, np.int32)])
        self.shape_ref = np.concatenate([self.shape_ref, np.full(nf, mesh_id, np.int32)])
        self.shape_face = np.concatenate([self.shape_face, np.arange(nf, dtype=np.int32)])
        self.shape_area_light = np.concatenate([self.shape_area_light, al])
        return mesh_id

    def add_sphere(self, center, radius, material_id, emission=None):
        sid = len(self.spheres)
        self.spheres.append(Sphere(tuple(center), float(radius), int(material_id)))
        al = -1
        if emission is not None:
            al = len(self.lights)
            self.lights.append(Light(1, self.n_shapes, tuple(emission)))
        self.shape_kind = np.concatenate([self.shape_kind, np.zeros(1, np.int32)])
        self.shape_ref = np.concatenate([self.shape_ref, np.full(1, sid, np.int32)])
        self.shape_face = np.concatenate([self.shape_face, np.zeros(1, np.int32)])
        self.shape_area_light = np.concatenate([self.shape_area_light, np.full(1, al, np.int32)])
        return sid

    def add_prototype(self, positions, indices, material_id, normals=None, uvs=None):
        """a mesh that is only placed through add_instance: it gets no shapes of its own"""
        self.meshes.append(Mesh(np.ascontiguousarray(positions, np.float64).reshape(-1, 3),
                                np.ascontiguousarray(indices, np.int32).reshape(-1, 3), int(material_id),
                                None if normals is None else np.ascontiguousarray(normals, np.float64).reshape(-1, 3),
                                None if uvs is None else np.ascontiguousarray(uvs, np.float64).reshape(-1, 2)))
        return len(self.meshes) - 1

    def add_instance(self, mesh_id, xform, material_id=-1):
        """EXTENSION (TakeInstance): place mesh `mesh_id` under the 3x4 affine object->world transform `xform`"""
        x = np.ascontiguousarray(xform, np.float64).reshape(3, 4)
        self.instance_mesh.append(int(mesh_id))
        self.instance_material.append(int(material_id))
        self.instance_xform.append(x)
        return len(self.instance_mesh) - 1

    def flattened(self):
        """the same scene with every instance expanded to world-space triangles (one mesh per instance): what an
        instanced render is specified to equal, to fp rounding"""
        import copy

        out = copy.copy(self)
        out.meshes = list(self.meshes)
        out.shape_kind, out.shape_ref = self.shape_kind.copy(), self.shape_ref.copy()
        out.shape_face, out.shape_area_light = self.shape_face.copy(), self.shape_area_light.copy()
        out.lights = list(self.lights)
        out.instance_mesh, out.instance_material, out.instance_xform = [], [], []
        kinds, refs, faces, als = [out.shape_kind], [out.shape_ref], [out.shape_face], [out.shape_area_light]
        for mid, mat, x in zip(self.instance_mesh, self.instance_material, self.instance_xform):
            m = self.meshes[mid]
            # (elementwise, left to right: the arithmetic of the library's own flattening, TAKE_INSTANCES_FLATTEN —
            # the two expansions are bit-identical)
            px, py, pz = m.positions[:, 0], m.positions[:, 1], m.positions[:, 2]
            pos = np.stack([x[r, 0] * px + x[r, 1] * py + x[r, 2] * pz + x[r, 3] for r in range(3)], axis=1)
            nrm = None
            if m.normals is not None:
                # rows: (L^-T n)^T = n^T L^-1; NOT re-normalised per vertex — interpolation commutes with the linear
                # map only then (the interpolated normal is normalised at the hit, src/shape.cpp:105)
                (a00, a01, a02), (a10, a11, a12), (a20, a21, a22) = x[:, :3]
                det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20)
                inv = np.array([[(a11 * a22 - a12 * a21) / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det],
                                [(a12 * a20 - a10 * a22) / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det],
                                [(a10 * a21 - a11 * a20) / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det]])
                nx, ny, nz = m.normals[:, 0], m.normals[:, 1], m.normals[:, 2]
                nrm = np.stack([nx * inv[0, c] + ny * inv[1, c] + nz * inv[2, c] for c in range(3)], axis=1)
            # (add_mesh, with the four shape arrays concatenated once at the end: 1000 placements of 10k triangles)
            nf = m.indices.shape[0]
            out.meshes.append(Mesh(np.ascontiguousarray(pos, np.float64), m.indices, m.material_id if mat < 0 else mat,
                                   None if nrm is None else np.ascontiguousarray(nrm, np.float64), m.uvs))
            kinds.append(np.ones(nf, np.int32)), refs.append(np.full(nf, len(out.meshes) - 1, np.int32))
            faces.append(np.arange(nf, dtype=np.int32)), als.append(np.full(nf, -1, np.int32))
        out.shape_kind, out.shape_ref = np.concatenate(kinds), np.concatenate(refs)
        out.shape_face, out.shape_area_light = np.concatenate(faces), np.concatenate(als)
        return out

    def add_envmap(self, image, scale=(1.0, 1.0, 1.0)):
        """Environment-map light (EXTENSION, TakeLight kind 2 — the reference has only `background`): an
        equirectangular (h, w, 3) radiance image, y up, row 0 = zenith; importance-sampled by luminance * sin(theta)
        and seen by rays that leave the scene.  It counts as one more light in the uniform light pick."""
        a = np.ascontiguousarray(image, np.float64)
        assert a.ndim == 3 and a.shape[2] == 3
        self.images.append(a)
        self.lights.append(Light(2, len(self.images) - 1, tuple(float(x) for x in scale)))
        return len(self.lights) - 1

    def add_material(self, tag, color=(0.5, 0.5, 0.5), param=(0.0, 0.0, 0.0, 0.0), tex_image=None,
                     uvxf=(1.0, 1.0, 0.0, 0.0)):
        p = tuple(float(x) for x in param) + (0.0,) * (12 - len(param))
        self.materials.append(Material(int(tag), tuple(color), 0 if tex_image is None else 1,
                                       0 if tex_image is None else int(tex_image), tuple(uvxf), p))
        return len(self.materials) - 1

    # -- C view -------------------------------------------------------------------------------------------
    def to_desc(self):
        """-> (TakeSceneDesc, keepalive).  Arrays are referenced in place: keep `keepalive` (and self)
        alive until the callee returns."""
        keep = []

        def dptr(a):
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))

        def iptr(a):
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_int32))

        d = D.TakeSceneDesc()
        d.camera.width, d.camera.height = self.width, self.height
        d.camera.lookfrom = D.c_double3(*self.lookfrom)
        d.camera.lookat = D.c_double3(*self.lookat)
        d.camera.up = D.c_double3(*self.up)
        d.camera.vfov = self.vfov
        d.background = D.c_double3(*self.background)
        meshes = (D.TakeMesh * max(len(self.meshes), 1))()
        for i, m in enumerate(self.meshes):
            if hasattr(m, "c"):  # capi.DeviceMesh: arrays already in device memory (decoded there from a PLY file)
                meshes[i] = m.c
                meshes[i].material_id = m.material_id
                keep.append(m)
                continue
            pos = np.ascontiguousarray(m.positions, np.float64)
            idx = np.ascontiguousarray(m.indices, np.int32)
            meshes[i].n_vertices = pos.shape[0]
            meshes[i].n_faces = idx.shape[0]
            meshes[i].positions = dptr(pos)
            meshes[i].indices = iptr(idx)
            if m.normals is not None:
                meshes[i].normals = dptr(np.ascontiguousarray(m.normals, np.float64))
            if m.uvs is not None:
                meshes[i].uvs = dptr(np.ascontiguousarray(m.uvs, np.float64))
            meshes[i].material_id = m.material_id
        spheres = (D.TakeSphere * max(len(self.spheres), 1))()
        for i, s in enumerate(self.spheres):
            spheres[i].center = D.c_double3(*s.center)
            spheres[i].radius = s.radius
            spheres[i].material_id = s.material_id
        lights = (D.TakeLight * max(len(self.lights), 1))()
        for i, l in enumerate(self.lights):
            lights[i].kind, lights[i].shape_id = l.kind, l.shape_id
            lights[i].intensity = D.c_double3(*l.intensity)
            lights[i].position = D.c_double3(*l.position)
        mats = (D.TakeMaterial * max(len(self.materials), 1))()
        for i, m in enumerate(self.materials):
            mats[i].tag = m.tag
            mats[i].reflectance.kind = m.tex_kind
            mats[i].reflectance.image_id = m.tex_image
            mats[i].reflectance.value = D.c_double3(*m.color)
            (mats[i].reflectance.uscale, mats[i].reflectance.vscale, mats[i].reflectance.uoffset,
             mats[i].reflectance.voffset) = m.uvxf
            mats[i].param = (C.c_double * 12)(*(tuple(m.param) + (0.0,) * (12 - len(m.param))))
        images = (D.TakeImage3 * max(len(self.images), 1))()
        for i, im in enumerate(self.images):
            a = np.ascontiguousarray(im, np.float64)
            images[i].height, images[i].width = a.shape[0], a.shape[1]
            images[i].data = dptr(a)
        keep += [meshes, spheres, lights, mats, images]
        d.n_meshes, d.meshes = len(self.meshes), meshes
        d.n_spheres, d.spheres = len(self.spheres), spheres
        d.n_shapes = self.n_shapes
        d.shape_kind = iptr(np.ascontiguousarray(self.shape_kind, np.int32))
        d.shape_ref = iptr(np.ascontiguousarray(self.shape_ref, np.int32))
        d.shape_face = iptr(np.ascontiguousarray(self.shape_face, np.int32))
        d.shape_area_light = iptr(np.ascontiguousarray(self.shape_area_light, np.int32))
        d.n_lights, d.lights = len(self.lights), lights
        d.n_materials, d.materials = len(self.materials), mats
        d.n_images, d.images = len(self.images), images
        inst = (D.TakeInstance * max(len(self.instance_mesh), 1))()
        for i, (mid, mat, x) in enumerate(zip(self.instance_mesh, self.instance_material, self.instance_xform)):
            inst[i].mesh_id, inst[i].material_id = mid, mat
            inst[i].xform = (C.c_double * 12)(*x.reshape(-1))
        keep.append(inst)
        d.n_instances, d.instances = len(self.instance_mesh), inst
        return d, keep


# ------------------------------------------------------------------------------------------- .tkscene I/O
class _Reader:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, n):
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o).copy()
        self.o += a.nbytes
        return a


def load_tkscene(path) -> SceneData:
    with open(path, "rb") as f:
        r = _Reader(f.read())
    if r.b[:8] != b"TKSCENE1":
        raise ValueError(f"{path}: not a .tkscene file")
    r.o = 8
    s = SceneData()
    s.width, s.height = r.take("ii")
    s.lookfrom, s.lookat, s.up = tuple(r.arr("<f8", 3)), tuple(r.arr("<f8", 3)), tuple(r.arr("<f8", 3))
    s.vfov = r.take("d")
    s.background = tuple(r.arr("<f8", 3))
    s.spp, s.max_depth = r.take("ii")
    for _ in range(r.take("i")):
        nv, nf = r.take("qq")
        mat, has_n, has_uv, _pad = r.take("iiii")
        pos = r.arr("<f8", nv * 3).reshape(nv, 3)
        idx = r.arr("<i4", nf * 3).reshape(nf, 3)
        nrm = r.arr("<f8", nv * 3).reshape(nv, 3) if has_n else None
        uv = r.arr("<f8", nv * 2).reshape(nv, 2) if has_uv else None
        s.meshes.append(Mesh(pos, idx, mat, nrm, uv))
    for _ in range(r.take("i")):
        c = tuple(r.arr("<f8", 3))
        rad = r.take("d")
        mat, _pad = r.take("ii")
        s.spheres.append(Sphere(c, rad, mat))
    n = r.take("q")
    s.shape_kind, s.shape_ref = r.arr("<i4", n), r.arr("<i4", n)
    s.shape_face, s.shape_area_light = r.arr("<i4", n), r.arr("<i4", n)
    for _ in range(r.take("i")):
        kind, sid = r.take("ii")
        s.lights.append(Light(kind, sid, tuple(r.arr("<f8", 3)), tuple(r.arr("<f8", 3))))
    for _ in range(r.take("i")):
        tag, tk, ti, n_more = r.take("iiii")
        col = tuple(r.arr("<f8", 3))
        uvxf = tuple(r.arr("<f8", 4))
        par = tuple(r.arr("<f8", 4 + n_more)) + (0.0,) * (8 - n_more)
        s.materials.append(Material(tag, col, tk, ti, uvxf, par))
    for _ in range(r.take("i")):
        w, h = r.take("ii")
        s.images.append(r.arr("<f8", w * h * 3).reshape(h, w, 3))
    return s


def save_tkscene(path, s: SceneData):
    with open(path, "wb") as f:
        f.write(b"TKSCENE1")
        f.write(struct.pack("<ii", s.width, s.height))
        for v in (s.lookfrom, s.lookat, s.up):
            f.write(np.asarray(v, "<f8").tobytes())
        f.write(struct.pack("<d", s.vfov))
        f.write(np.asarray(s.background, "<f8").tobytes())
        f.write(struct.pack("<ii", s.spp, s.max_depth))
        f.write(struct.pack("<i", len(s.meshes)))
        for m in s.meshes:
            f.write(struct.pack("<qqiiii", m.positions.shape[0], m.indices.shape[0], m.material_id,
                                int(m.normals is not None), int(m.uvs is not None), 0))
            f.write(np.asarray(m.positions, "<f8").tobytes())
            f.write(np.asarray(m.indices, "<i4").tobytes())
            if m.normals is not None:
                f.write(np.asarray(m.normals, "<f8").tobytes())
            if m.uvs is not None:
                f.write(np.asarray(m.uvs, "<f8").tobytes())
        f.write(struct.pack("<i", len(s.spheres)))
        for sp in s.spheres:
            f.write(np.asarray(sp.center, "<f8").tobytes())
            f.write(struct.pack("<dii", sp.radius, sp.material_id, 0))
        f.write(struct.pack("<q", s.n_shapes))
        for a in (s.shape_kind, s.shape_ref, s.shape_face, s.shape_area_light):
            f.write(np.asarray(a, "<i4").tobytes())
        f.write(struct.pack("<i", len(s.lights)))
        for l in s.lights:
            f.write(struct.pack("<ii", l.kind, l.shape_id))
            f.write(np.asarray(l.intensity, "<f8").tobytes())
            f.write(np.asarray(l.position, "<f8").tobytes())
        f.write(struct.pack("<i", len(s.materials)))
        for m in s.materials:
            par = tuple(m.param) + (0.0,) * (12 - len(m.param))
            n_more = 8 if any(par[4:]) else 0  # scenes without Disney parameters keep the first format's bytes
            f.write(struct.pack("<iiii", m.tag, m.tex_kind, m.tex_image, n_more))
            f.write(np.asarray(m.color, "<f8").tobytes())
            f.write(np.asarray(m.uvxf, "<f8").tobytes())
            f.write(np.asarray(par[:4 + n_more], "<f8").tobytes())
        f.write(struct.pack("<i", len(s.images)))
        for im in s.images:
            f.write(struct.pack("<ii", im.shape[1], im.shape[0]))
            f.write(np.asarray(im, "<f8").tobytes())
