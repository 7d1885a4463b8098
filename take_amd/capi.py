"""ctypes binding of libtake_hip.so (include/take_hip.h).

The library is the product; this module only marshals.  There is no fallback: if the shared object is
missing or no HIP device is visible every call raises `TakeError`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import cdefs as D

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAKE_HIP_LIB") or os.path.join(_PKG, "libtake_hip.so")  # TAKE_HIP_LIB: tuning builds
_LIB = None

EXPORTS = [
    "take_hip_last_error", "take_hip_abi_version", "take_hip_device_count", "take_hip_scene_create",
    "take_hip_scene_destroy", "take_hip_render", "take_hip_render_device", "take_hip_render_rows",
    "take_hip_render_accumulate", "take_hip_accumulated_samples",
    "take_hip_trace_closest", "take_hip_trace_any", "take_hip_trace_closest_device", "take_hip_get_counters",
    "take_hip_set_instrumentation", "take_hip_scene_stats", "take_hip_debug_table",
    "take_hip_group_create", "take_hip_group_destroy", "take_hip_group_render", "take_hip_group_render_device",
    "take_hip_group_size", "take_hip_group_get_counters", "take_hip_pack_exr_scanlines", "take_hip_render_exr_scanlines",
    "take_hip_ply_layout", "take_hip_mesh_from_ply", "take_hip_mesh_from_ply_file",
    "take_hip_mesh_from_serialized", "take_hip_mesh_from_serialized_file", "take_hip_mesh_download", "take_hip_mesh_release",
]


class TakeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"take_hip error {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile libtake_hip.so for gfx950 with hipcc (take_amd/csrc/Makefile).  Cross-compiles without a GPU."""
    src = os.path.join(_PKG, "csrc")
    args = ["make", "-C", src]
    if force:
        args.append("-B")
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libtake_hip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TakeError(-3, f"{LIB_PATH} not built (run __graft_entry__.build() / make -C take_amd/csrc)")
        L = C.CDLL(LIB_PATH)
        L.take_hip_last_error.restype = C.c_char_p
        L.take_hip_scene_create.argtypes = [C.POINTER(D.TakeSceneDesc), C.POINTER(D.TakeBuildOpts),
                                            C.POINTER(C.c_void_p)]
        L.take_hip_scene_destroy.argtypes = [C.c_void_p]
        L.take_hip_render.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_void_p]
        L.take_hip_render_device.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_void_p, C.c_void_p]
        L.take_hip_render_accumulate.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_int32, C.c_void_p, C.c_void_p]
        L.take_hip_accumulated_samples.argtypes = [C.c_void_p]
        L.take_hip_accumulated_samples.restype = C.c_int64
        L.take_hip_render_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.take_hip_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.take_hip_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
        L.take_hip_trace_closest_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                                    C.c_void_p]
        L.take_hip_get_counters.argtypes = [C.c_void_p, C.POINTER(D.TakeCounters)]
        L.take_hip_set_instrumentation.argtypes = [C.c_void_p, C.c_int32]
        L.take_hip_scene_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.take_hip_pack_exr_scanlines.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.take_hip_render_exr_scanlines.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_void_p]
        L.take_hip_group_create.argtypes = [C.POINTER(D.TakeSceneDesc), C.POINTER(D.TakeBuildOpts), C.c_int32,
                                            C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.take_hip_group_destroy.argtypes = [C.c_void_p]
        L.take_hip_group_render.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_void_p]
        L.take_hip_group_render_device.argtypes = [C.c_void_p, C.POINTER(D.TakeRenderOpts), C.c_void_p]
        L.take_hip_group_size.argtypes = [C.c_void_p]
        L.take_hip_group_get_counters.argtypes = [C.c_void_p, C.c_int32, C.POINTER(D.TakeCounters)]
        L.take_hip_ply_layout.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(D.TakePlyLayout)]
        L.take_hip_mesh_from_ply.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.TakeMesh)]
        L.take_hip_mesh_from_ply_file.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.TakeMesh)]
        L.take_hip_mesh_from_serialized.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.TakeMesh)]
        L.take_hip_mesh_from_serialized_file.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(D.TakeMesh)]
        L.take_hip_mesh_download.argtypes = [C.POINTER(D.TakeMesh), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.take_hip_mesh_release.argtypes = [C.POINTER(D.TakeMesh)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc < 0:
        raise TakeError(rc, lib().take_hip_last_error().decode())
    return rc


def device_count():
    return _check(lib().take_hip_device_count())


def ply_layout(data):
    """what the header of a binary PLY file says about its vertex / face elements (host only, no GPU)"""
    buf = bytes(data)
    out = D.TakePlyLayout()
    _check(lib().take_hip_ply_layout(buf, len(buf), C.byref(out)))
    return {k: getattr(out, k) for k, _ in D.TakePlyLayout._fields_ if k != "reserved"}


class DeviceMesh:
    """A triangle mesh decoded from a binary PLY file or a Mitsuba `.serialized` file ON the device
    (take_hip_mesh_from_ply / _from_serialized: replace the reference's parse_ply, src/parse/parse_ply.cpp:9-123, and
    parse_serialized, src/parse/parse_serialized.cpp:174-256).  The arrays are device memory owned by the library; put the
    object into SceneData.meshes like a scene.Mesh.  `source`: a path or the file's bytes; the format is told from the
    first bytes (`ply` / anything else = serialized, whose sub-mesh `shape_index` picks).  to_world: 4x4 (the
    reference's Matrix4x4); inv_to_world: the caller's inverse of it (the reference passes its own
    `inverse(to_world)`), default numpy's."""

    def __init__(self, source, material_id=0, to_world=None, inv_to_world=None, shape_index=0):
        self.c = D.TakeMesh()
        xw = xi = None
        if to_world is not None:
            xw = np.ascontiguousarray(to_world, np.float64).reshape(4, 4)
            xi = np.ascontiguousarray(np.linalg.inv(xw) if inv_to_world is None else inv_to_world, np.float64).reshape(4, 4)
        a = None if xw is None else xw.ctypes.data
        b = None if xi is None else xi.ctypes.data
        if isinstance(source, (bytes, bytearray, memoryview)):
            buf = bytes(source)
            if buf[:3] == b"ply":
                _check(lib().take_hip_mesh_from_ply(buf, len(buf), a, b, int(material_id), C.byref(self.c)))
            else:
                _check(lib().take_hip_mesh_from_serialized(buf, len(buf), int(shape_index), a, b, int(material_id), C.byref(self.c)))
        else:
            with open(source, "rb") as f:
                is_ply = f.read(3) == b"ply"
            if is_ply:
                _check(lib().take_hip_mesh_from_ply_file(os.fsencode(source), a, b, int(material_id), C.byref(self.c)))
            else:
                _check(lib().take_hip_mesh_from_serialized_file(os.fsencode(source), int(shape_index), a, b, int(material_id), C.byref(self.c)))
        self.material_id = int(material_id)

    n_vertices = property(lambda self: int(self.c.n_vertices))
    n_faces = property(lambda self: int(self.c.n_faces))

    def download(self):
        """-> scene.Mesh with host copies of the arrays (tests; the render path never needs it)"""
        from .scene import Mesh

        nv, nf = self.n_vertices, self.n_faces
        pos, idx = np.zeros((nv, 3), np.float64), np.zeros((nf, 3), np.int32)
        nrm = np.zeros((nv, 3), np.float64) if self.c.normals else None
        uv = np.zeros((nv, 2), np.float64) if self.c.uvs else None
        _check(lib().take_hip_mesh_download(C.byref(self.c), pos.ctypes.data, idx.ctypes.data,
                                            None if nrm is None else nrm.ctypes.data, None if uv is None else uv.ctypes.data))
        return Mesh(pos, idx, self.material_id, nrm, uv)

    def close(self):
        if self.c.flags:
            lib().take_hip_mesh_release(C.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DEBUG_TABLES = {"material": (0, 27, 14), "light": (1, 30, 9), "texture": (2, 6, 3), "to_world": (3, 6, 3),
                "hemicos": (4, 1, 4), "burley": (5, 30, 14)}


def debug_table(name, inp, rnd, precision=D.TAKE_PRECISION_F64):
    """Device shading functions on golden-table rows (test hook).  rnd: (n, 8) random_real draws per row."""
    kind, cin, cout = DEBUG_TABLES[name]
    inp = np.ascontiguousarray(inp, np.float64).reshape(-1, cin)
    rnd = np.ascontiguousarray(rnd, np.float64).reshape(-1, 8)
    n = inp.shape[0]
    out = np.zeros((n, cout), np.float64)
    f = lib().take_hip_debug_table
    f.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    _check(f(kind, precision, inp.ctypes.data, n, cin, rnd.ctypes.data, out.ctypes.data, cout))
    return out


class Scene:
    """A scene resident on the current HIP device: flattened `Scene` + wide BVH in HBM."""

    def __init__(self, scene_data, precision=D.TAKE_PRECISION_F32, bvh_threads=0, max_leaf_size=0,
                 builder=D.TAKE_BUILDER_AUTO, burley_lobes=False, flatten_instances=False):
        """flatten_instances: placements (TakeInstance) are expanded to world-space triangles by scene_create instead of
        being traversed on two levels (TAKE_INSTANCES_FLATTEN).
        burley_lobes: the scene's Disney materials (tags 7..11: Lambert clones, as upstream) are rendered with the
        real lobes (tags 12..16, an extension — DESIGN.md §4d)"""
        self.sd = scene_data
        self.precision = precision
        self.dtype = np.float32 if precision == D.TAKE_PRECISION_F32 else np.float64
        desc, keep = scene_data.to_desc()
        opts = D.TakeBuildOpts(precision, bvh_threads, max_leaf_size, builder, 1 if burley_lobes else 0,
                               D.TAKE_INSTANCES_FLATTEN if flatten_instances else D.TAKE_INSTANCES_TWO_LEVEL)
        h = C.c_void_p()
        _check(lib().take_hip_scene_create(C.byref(desc), C.byref(opts), C.byref(h)))
        self.h = h
        del keep

    def close(self):
        if getattr(self, "h", None):
            lib().take_hip_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, spp, max_depth, seed, ray_epsilon, strip_first, strip_stride, samples_per_batch, integrator=0):
        o = D.TakeRenderOpts()
        o.exact_bounces = int(getattr(self, "exact_bounces", 0))  # TAKE_PRECISION_MIXED scenes (0 = the library's default)
        o.spp, o.max_depth, o.seed, o.ray_epsilon = int(spp), int(max_depth), int(seed), float(ray_epsilon)
        o.strip_first, o.strip_stride, o.samples_per_batch = int(strip_first), int(strip_stride), int(samples_per_batch)
        o.integrator = int(integrator)
        return o

    def rows(self, strip_first=0, strip_stride=1):
        n = _check(lib().take_hip_render_rows(self.h, strip_first, strip_stride, None))
        out = (C.c_int32 * max(n, 1))()
        _check(lib().take_hip_render_rows(self.h, strip_first, strip_stride, out))
        return np.array(out[:n], np.int32)

    def render(self, spp=None, max_depth=None, seed=0, ray_epsilon=0.0, strip_first=0, strip_stride=1,
               samples_per_batch=0, integrator=0):
        """-> (rows, W, 3) image rows owned by this strip set, top row first (host array)."""
        spp = self.sd.spp if spp is None else spp
        max_depth = self.sd.max_depth if max_depth is None else max_depth
        o = self._opts(spp, max_depth, seed, ray_epsilon, strip_first, strip_stride, samples_per_batch, integrator)
        n = _check(lib().take_hip_render_rows(self.h, strip_first, strip_stride, None))
        out = np.zeros((n, self.sd.width, 3), self.dtype)
        _check(lib().take_hip_render(self.h, C.byref(o), out.ctypes.data))
        return out

    def render_device(self, d_ptr, spp, max_depth, seed=0, ray_epsilon=0.0, strip_first=0, strip_stride=1,
                      samples_per_batch=0, stream=None, integrator=0):
        """Render into device memory at `d_ptr` (e.g. a torch tensor's data_ptr()); blocks until done."""
        o = self._opts(spp, max_depth, seed, ray_epsilon, strip_first, strip_stride, samples_per_batch, integrator)
        _check(lib().take_hip_render_device(self.h, C.byref(o), C.c_void_p(d_ptr), C.c_void_p(stream or 0)))

    def render_exr_scanlines(self, spp=None, max_depth=None, seed=0, samples_per_batch=0, integrator=0):
        """Render and convert on the device: -> uint16 (H, 3, W), per scanline the B, G, R halves of the reference's
        image.exr (take_amd.exr.write_exr_scanlines frames them into the file)."""
        spp = self.sd.spp if spp is None else spp
        max_depth = self.sd.max_depth if max_depth is None else max_depth
        o = self._opts(spp, max_depth, seed, 0.0, 0, 1, samples_per_batch, integrator)
        out = np.zeros((self.sd.height, 3, self.sd.width), np.uint16)
        _check(lib().take_hip_render_exr_scanlines(self.h, C.byref(o), out.ctypes.data))
        return out

    def render_accumulate(self, d_ptr, more_spp, max_depth, seed=0, restart=False, ray_epsilon=0.0, strip_first=0,
                          strip_stride=1, samples_per_batch=0, stream=None, integrator=0):
        """progressive rendering: `more_spp` further samples per pixel, mean over all samples so far -> device buffer"""
        o = self._opts(more_spp, max_depth, seed, ray_epsilon, strip_first, strip_stride, samples_per_batch, integrator)
        _check(lib().take_hip_render_accumulate(self.h, C.byref(o), 1 if restart else 0, C.c_void_p(d_ptr), C.c_void_p(stream or 0)))
        return int(lib().take_hip_accumulated_samples(self.h))

    def trace_closest(self, rays_abi):
        """rays_abi: (n,8) array in TakeRayF/D layout (org3 tmin dir3 tmax) -> structured hits"""
        rays = np.ascontiguousarray(rays_abi, self.dtype)
        n = rays.shape[0]
        if self.precision != D.TAKE_PRECISION_F32:
            hits = np.zeros(n, dtype=[("shape_id", "<i4"), ("reserved", "<i4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8")])
        else:
            hits = np.zeros(n, dtype=[("shape_id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
        _check(lib().take_hip_trace_closest(self.h, rays.ctypes.data, n, hits.ctypes.data))
        return hits

    def trace_any(self, rays_abi):
        rays = np.ascontiguousarray(rays_abi, self.dtype)
        occ = np.zeros(rays.shape[0], np.int32)
        _check(lib().take_hip_trace_any(self.h, rays.ctypes.data, rays.shape[0],
                                        occ.ctypes.data_as(C.POINTER(C.c_int32))))
        return occ

    def trace_closest_device(self, d_rays, n, d_hits, count_mode=False, stream=None):
        _check(lib().take_hip_trace_closest_device(self.h, C.c_void_p(d_rays), int(n), C.c_void_p(d_hits),
                                                   int(count_mode), C.c_void_p(stream or 0)))

    def set_instrumentation(self, timing=False, counting=False):
        _check(lib().take_hip_set_instrumentation(self.h, (1 if timing else 0) | (2 if counting else 0)))

    def counters(self):
        c = D.TakeCounters()
        _check(lib().take_hip_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def stats(self):
        nn, npr, dep, by = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64()
        _check(lib().take_hip_scene_stats(self.h, C.byref(nn), C.byref(npr), C.byref(dep), C.byref(by)))
        return {"n_nodes": nn.value, "n_prims": npr.value, "depth": dep.value, "device_bytes": by.value}


class SceneGroup:
    """The scene replicated on several GPUs of ONE process (take_hip_group_*): what the reference's single-process
    C++ host uses in place of its thread pool.  `devices`: HIP device per shard; a device may repeat (logical shards)."""

    def __init__(self, scene_data, devices, precision=D.TAKE_PRECISION_F32, bvh_threads=0, max_leaf_size=0,
                 builder=D.TAKE_BUILDER_AUTO, burley_lobes=False, flatten_instances=False):
        self.sd = scene_data
        self.precision = precision
        self.dtype = np.float32 if precision == D.TAKE_PRECISION_F32 else np.float64
        desc, keep = scene_data.to_desc()
        opts = D.TakeBuildOpts(precision, bvh_threads, max_leaf_size, builder, 1 if burley_lobes else 0,
                               D.TAKE_INSTANCES_FLATTEN if flatten_instances else D.TAKE_INSTANCES_TWO_LEVEL)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib().take_hip_group_create(C.byref(desc), C.byref(opts), len(devices), devs, C.byref(h)))
        self.h = h
        del keep

    def close(self):
        if getattr(self, "h", None):
            lib().take_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return _check(lib().take_hip_group_size(self.h))

    def render(self, spp=None, max_depth=None, seed=0, samples_per_batch=0, integrator=0):
        """-> (H, W, 3) host image assembled on the group's first device"""
        o = D.TakeRenderOpts()
        o.spp = int(self.sd.spp if spp is None else spp)
        o.max_depth = int(self.sd.max_depth if max_depth is None else max_depth)
        o.seed, o.samples_per_batch, o.integrator = int(seed), int(samples_per_batch), int(integrator)
        out = np.zeros((self.sd.height, self.sd.width, 3), self.dtype)
        _check(lib().take_hip_group_render(self.h, C.byref(o), out.ctypes.data))
        return out

    def counters(self, k):
        c = D.TakeCounters()
        _check(lib().take_hip_group_get_counters(self.h, int(k), C.byref(c)))
        return c.as_dict()
