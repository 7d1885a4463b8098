"""TEST INFRASTRUCTURE — ctypes front end of oracle/libtake_oracle.so (the CPU restatement of the
reference, oracle/take_oracle.hpp).  Importable only from tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg; the product package (take_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libtake_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("take_oracle.cpp", "take_oracle.hpp")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "port"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        for name in ("random_real", "slab", "tri", "sphere", "to_world", "hemicos", "material", "texture", "light",
                     "bvh", "burley"):
            fn = getattr(L, "oracle_tab_" + name)
            fn.argtypes = [dp, C.c_int64, dp]
            fn.restype = None
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_isect.argtypes = [C.c_void_p, dp, C.c_int64, dp]
        L.oracle_isect_brute.argtypes = [C.c_void_p, dp, C.c_int64, dp]
        L.oracle_pt_mt.argtypes = [C.c_void_p, C.c_int, dp, C.c_int64, dp]
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, dp, C.c_int]
        L.oracle_render.restype = C.c_double
        L.oracle_pt_mt2.argtypes = [C.c_void_p, C.c_int, dp, C.c_int64, dp, C.c_int]
        L.oracle_render2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, dp, C.c_int, C.c_int]
        L.oracle_render2.restype = C.c_double
        L.oracle_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.oracle_counter_words.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


TABLE_COLS = {"random_real": (1, 64), "slab": (14, 1), "tri": (34, 15), "sphere": (12, 15), "to_world": (6, 3),
              "hemicos": (1, 4), "material": (27, 14), "texture": (6, 3), "light": (30, 9),
              "burley": (30, 14)}


def table(name, inp):
    """Run the restatement of one per-function table (same columns as oracle/ref_harness.cpp)."""
    inp = np.ascontiguousarray(inp, np.float64)
    if name == "bvh":
        n = inp.size // 6
        out = np.zeros(2 + 9 * (2 * n - 1), np.float64)
        lib().oracle_tab_bvh(_dp(inp), n, _dp(out))
        return out
    cin, cout = TABLE_COLS[name]
    n = inp.size // cin
    out = np.zeros(n * cout, np.float64)
    getattr(lib(), "oracle_tab_" + name)(_dp(inp), n, _dp(out))
    return out.reshape(n, cout)


RNG_MT_PER_TILE, RNG_COUNTER = 0, 1


class OracleScene:
    """A scene in the restatement: reference BVH (median split), reference traversal."""

    def __init__(self, scene_data, precision=1, ray_eps=0.0):
        self.sd = scene_data
        desc, keep = scene_data.to_desc()
        self.h = lib().oracle_scene_create(C.addressof(desc), int(precision), float(ray_eps))
        del keep
        self.precision = precision

    def close(self):
        if self.h:
            lib().oracle_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def isect(self, rays):
        """rays (n,8): org dir tmin tmax -> (n,19): hit t pos3 gn3 sn3 uv2 mat light | occluded shape_id bu bv"""
        rays = np.ascontiguousarray(rays, np.float64)
        out = np.zeros((rays.shape[0], 19), np.float64)
        lib().oracle_isect(self.h, _dp(rays), rays.shape[0], _dp(out))
        return out

    def isect_brute(self, rays):
        rays = np.ascontiguousarray(rays, np.float64)
        out = np.zeros((rays.shape[0], 4), np.float64)
        lib().oracle_isect_brute(self.h, _dp(rays), rays.shape[0], _dp(out))
        return out

    def pt_mt(self, max_depth, inp, integrator=0):
        """one integrator call per row (org3 dir3 seed): radiance3 + the next random_real; `integrator` as in
        TakeRenderOpts (0 path_tracing, 1 raw, 2 one-sample MIS, 3 one-sample MIS with power-based light picking)"""
        inp = np.ascontiguousarray(inp, np.float64)
        out = np.zeros((inp.shape[0], 4), np.float64)
        lib().oracle_pt_mt2(self.h, int(max_depth), _dp(inp), inp.shape[0], _dp(out), int(integrator))
        return out

    def render(self, spp, max_depth, rng_mode=RNG_COUNTER, seed=0, threads=None, counters=False, integrator=0):
        """-> (H,W,3) float64 image (row 0 = top); self.seconds = tile-loop time"""
        threads = threads or os.cpu_count() or 1
        out = np.zeros((self.sd.height, self.sd.width, 3), np.float64)
        self.seconds = lib().oracle_render2(self.h, int(spp), int(max_depth), int(rng_mode), int(seed), int(threads),
                                            _dp(out), int(counters), int(integrator))
        return out

    def counters(self):
        c = (C.c_uint64 * 9)()
        lib().oracle_get_counters(self.h, c)
        keys = ["closest_rays", "closest_node_visits", "closest_box_tests", "closest_prim_tests", "shadow_rays",
                "shadow_node_visits", "shadow_box_tests", "shadow_prim_tests", "bounces"]
        return dict(zip(keys, [int(x) for x in c]))


def counter_words(seed, pixel, sample, n):
    out = (C.c_uint64 * n)()
    lib().oracle_counter_words(seed, pixel, sample, n, out)
    return np.array(out[:], dtype=np.uint64)
