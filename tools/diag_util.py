"""diagnostic: traversal work counters of one counting-mode render (quad utilisation of the trace kernel)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from take_amd import capi, scenes
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sd = scenes.soup_scene(tris, 1920, 1080, spp=1)
sc = capi.Scene(sd)
sc.set_instrumentation(timing=True, counting=True)
sc.render(spp=1, max_depth=50, seed=0)
c = sc.counters()
rays = c["rays_closest"] + c["rays_shadow"]
print({k: c[k] for k in ("rays_closest", "rays_shadow", "node_visits", "leaf_visits", "prim_tests", "wave_node_steps", "wave_leaf_steps")})
print(f"nodes/ray {c['node_visits']/rays:.1f} leaves/ray {c['leaf_visits']/rays:.1f} prims/ray {c['prim_tests']/rays:.1f}")
print(f"node-phase quad utilisation {c['node_visits']/(16*c['wave_node_steps']):.3f}  leaf-phase {c['leaf_visits']/(16*c['wave_leaf_steps']):.3f}")
print(f"wave steps per ray: node {c['wave_node_steps']*16/rays:.1f} leaf {c['wave_leaf_steps']*16/rays:.1f}")
