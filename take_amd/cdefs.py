"""ctypes mirror of include/take_hip.h (field order and types must match the header)."""
import ctypes as C

c_double3 = C.c_double * 3
c_double4 = C.c_double * 4
c_float3 = C.c_float * 3

TAKE_PRECISION_F32 = 0
TAKE_PRECISION_F64 = 1
TAKE_PRECISION_MIXED = 2  # first exact_bounces rounds in f64, the rest in f32; records and images are f64
TAKE_OK, TAKE_E_INVALID, TAKE_E_DEVICE, TAKE_E_NO_GPU, TAKE_E_NOMEM = 0, -1, -2, -3, -4

MAT_DIFFUSE, MAT_MIRROR, MAT_PLASTIC, MAT_PHONG, MAT_BLINN_PHONG, MAT_BLINN_PHONG_MICROFACET = range(6)
MAT_DISNEY_DIFFUSE, MAT_DISNEY_METAL, MAT_DISNEY_GLASS, MAT_DISNEY_CLEARCOAT, MAT_DISNEY_SHEEN, MAT_DISNEY_BSDF = range(6, 12)
# extension (tags of the real lobes; include/take_hip.h): the Disney tag + 5
MAT_BURLEY_METAL, MAT_BURLEY_GLASS, MAT_BURLEY_CLEARCOAT, MAT_BURLEY_SHEEN, MAT_BURLEY_BSDF = range(12, 17)
MATERIAL_PARAMS = 12


class TakeTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image_id", C.c_int32), ("value", c_double3),
                ("uscale", C.c_double), ("vscale", C.c_double), ("uoffset", C.c_double), ("voffset", C.c_double)]


class TakeMaterial(C.Structure):
    _fields_ = [("tag", C.c_int32), ("reserved", C.c_int32), ("reflectance", TakeTexture), ("param", C.c_double * 12)]


class TakeImage3(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.POINTER(C.c_double))]


class TakeMesh(C.Structure):
    _fields_ = [("n_vertices", C.c_int64), ("n_faces", C.c_int64), ("positions", C.POINTER(C.c_double)),
                ("indices", C.POINTER(C.c_int32)), ("normals", C.POINTER(C.c_double)), ("uvs", C.POINTER(C.c_double)),
                ("material_id", C.c_int32), ("flags", C.c_int32)]


TAKE_MESH_DEVICE_ARRAYS = 1


class TakePlyLayout(C.Structure):
    _fields_ = [("n_vertices", C.c_int64), ("n_faces", C.c_int64), ("vertex_offset", C.c_int64), ("face_offset", C.c_int64),
                ("vertex_stride", C.c_int32), ("face_stride", C.c_int32), ("has_normals", C.c_int32), ("has_uvs", C.c_int32),
                ("position_is_f64", C.c_int32), ("index_bytes", C.c_int32), ("header_bytes", C.c_int32), ("reserved", C.c_int32)]


class TakeSphere(C.Structure):
    _fields_ = [("center", c_double3), ("radius", C.c_double), ("material_id", C.c_int32), ("reserved", C.c_int32)]


class TakeLight(C.Structure):
    _fields_ = [("kind", C.c_int32), ("shape_id", C.c_int32), ("intensity", c_double3), ("position", c_double3)]


class TakeInstance(C.Structure):
    _fields_ = [("mesh_id", C.c_int32), ("material_id", C.c_int32), ("xform", C.c_double * 12)]


class TakeCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("lookfrom", c_double3), ("lookat", c_double3),
                ("up", c_double3), ("vfov", C.c_double)]


class TakeSceneDesc(C.Structure):
    _fields_ = [("camera", TakeCamera), ("background", c_double3),
                ("n_meshes", C.c_int32), ("n_spheres", C.c_int32),
                ("meshes", C.POINTER(TakeMesh)), ("spheres", C.POINTER(TakeSphere)),
                ("n_shapes", C.c_int64),
                ("shape_kind", C.POINTER(C.c_int32)), ("shape_ref", C.POINTER(C.c_int32)),
                ("shape_face", C.POINTER(C.c_int32)), ("shape_area_light", C.POINTER(C.c_int32)),
                ("n_lights", C.c_int32), ("n_materials", C.c_int32),
                ("lights", C.POINTER(TakeLight)), ("materials", C.POINTER(TakeMaterial)),
                ("n_images", C.c_int32), ("reserved", C.c_int32), ("images", C.POINTER(TakeImage3)),
                ("n_instances", C.c_int64), ("instances", C.POINTER(TakeInstance))]


class TakeBuildOpts(C.Structure):
    _fields_ = [("precision", C.c_int32), ("bvh_threads", C.c_int32), ("max_leaf_size", C.c_int32),
                ("builder", C.c_int32), ("burley_lobes", C.c_int32), ("instances", C.c_int32)]


TAKE_INSTANCES_TWO_LEVEL = 0    # placements are leaves of a top-level BVH
TAKE_INSTANCES_FLATTEN = 1      # placements expanded to world-space triangles by scene_create
TAKE_BUILDER_AUTO = 0          # host SAH below 4M shapes, device LBVH from there on
TAKE_BUILDER_DEVICE_LBVH = 1
TAKE_BUILDER_HOST_SAH = 2


class TakeRenderOpts(C.Structure):
    _fields_ = [("spp", C.c_int32), ("max_depth", C.c_int32), ("seed", C.c_uint64), ("ray_epsilon", C.c_double),
                ("strip_first", C.c_int32), ("strip_stride", C.c_int32), ("samples_per_batch", C.c_int32),
                ("integrator", C.c_int32), ("exact_bounces", C.c_int32), ("reserved", C.c_int32)]


# TakeRenderOpts.integrator: the reference's integrators (src/integrator/path_tracing.h:5, :114, :161, :274)
INTEGRATOR_PATH_MIS, INTEGRATOR_RAW, INTEGRATOR_ONE_SAMPLE_MIS, INTEGRATOR_ONE_SAMPLE_MIS_POWER = 0, 1, 2, 3


class TakeRayF(C.Structure):
    _fields_ = [("org", c_float3), ("tmin", C.c_float), ("dir", c_float3), ("tmax", C.c_float)]


class TakeRayD(C.Structure):
    _fields_ = [("org", c_double3), ("tmin", C.c_double), ("dir", c_double3), ("tmax", C.c_double)]


class TakeHitF(C.Structure):
    _fields_ = [("shape_id", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


class TakeHitD(C.Structure):
    _fields_ = [("shape_id", C.c_int32), ("reserved", C.c_int32), ("t", C.c_double), ("u", C.c_double),
                ("v", C.c_double)]


class TakeCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("bounces", C.c_uint64),
                ("ms_trace_closest", C.c_double), ("ms_trace_shadow", C.c_double), ("ms_shade", C.c_double),
                ("ms_other", C.c_double), ("ms_total", C.c_double),
                ("launches_trace_closest", C.c_uint64), ("launches_trace_shadow", C.c_uint64),
                ("node_bytes", C.c_uint64), ("prim_bytes", C.c_uint64), ("leaf_visits", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_leaf_steps", C.c_uint64),
                ("rays_closest_f32", C.c_uint64), ("ms_trace_closest_f32", C.c_double), ("launches_trace_closest_f32", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}
