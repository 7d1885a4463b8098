"""diagnostic: traversal work counters of one counting-mode render (ray-slot utilisation of the trace kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from take_amd import capi, scenes

PW = int(os.environ.get("TAKE_DIAG_SLOTS", "64"))  # ray slots per wave: 64 = one ray per lane (production), 32 = pair builds
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
precision = 1 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else 0
sd = scenes.soup_scene(tris, 1920, 1080, spp=spp, envmap=(2048, 1024))
sc = capi.Scene(sd, precision=precision)
print("precision", "f64" if precision else "f32", "tris", tris, "spp per batch", spp)
sc.set_instrumentation(timing=True, counting=True)
sc.render(spp=spp, max_depth=50, seed=0)
c = sc.counters()
rays = c["rays_closest"] + c["rays_shadow"]
print({k: c[k] for k in ("rays_closest", "rays_shadow", "node_visits", "leaf_visits", "prim_tests", "wave_node_steps",
                         "wave_leaf_steps")})
print("nodes/ray %.1f leaves/ray %.1f prims/ray %.1f" % (c["node_visits"] / rays, c["leaf_visits"] / rays,
                                                        c["prim_tests"] / rays))
print("node-phase utilisation %.3f  leaf-phase %.3f" % (c["node_visits"] / (PW * c["wave_node_steps"]),
                                                       c["leaf_visits"] / (PW * c["wave_leaf_steps"])))
print("wave steps per ray: node %.1f leaf %.1f   ms closest %.1f shadow %.1f" % (
    c["wave_node_steps"] * PW / rays, c["wave_leaf_steps"] * PW / rays, c["ms_trace_closest"], c["ms_trace_shadow"]))
