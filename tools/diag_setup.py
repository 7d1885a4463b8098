"""scene_create phases at 10M triangles (TAKE_HIP_VERBOSE): where the setup time goes, host SAH vs device LBVH"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TAKE_HIP_VERBOSE"] = "1"
from take_amd import capi, scenes
from take_amd import cdefs as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t = time.time(); sd = scenes.soup_scene(n, 640, 360, spp=1); print(f"scene generation {time.time()-t:.2f} s", flush=True)
for b, name in ((D.TAKE_BUILDER_DEVICE_LBVH, "device"), (D.TAKE_BUILDER_HOST_SAH, "host")):
    for rep in range(2):
        t = time.time(); sc = capi.Scene(sd, builder=b); dt = time.time() - t
        print(f"builder {name} rep {rep}: scene_create {dt:.3f} s  {sc.stats()}", flush=True)
        sc.close()
