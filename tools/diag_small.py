import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import golden_scene
from take_amd import capi, scenes
for name, sd, spp in (("cbox 256x256x16", golden_scene("cbox"), 16), ("soup100k 512x512x4", scenes.soup_scene(100_000, 512, 512, spp=4), 4)):
    if name.startswith("cbox"):
        sd.width = sd.height = 256
    sc = capi.Scene(sd)
    sc.render(spp=spp, max_depth=50, seed=0)
    sc.set_instrumentation(timing=True, counting=False)
    t = time.time()
    for _ in range(5):
        sc.render(spp=spp, max_depth=50, seed=0)
    wall = (time.time() - t) / 5 * 1e3
    c = sc.counters()
    print(f"{name}: wall {wall:.2f} ms per render, kernels {c['ms_trace_closest'] + c['ms_trace_shadow'] + c['ms_shade'] + c['ms_other']:.2f} ms (closest {c['ms_trace_closest']:.2f} shadow {c['ms_trace_shadow']:.2f} shade {c['ms_shade']:.2f} other {c['ms_other']:.2f}), total {c['ms_total']:.2f}")
    sc.close()
