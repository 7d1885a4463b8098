"""Experiment: do two renders on two streams overlap (trace of one with shade of the other)?
Two scene handles on one GPU, two host threads, each rendering half of the samples on its own stream, against one
render of all samples.  TAKE_HIP_TRACE_BLOCKS=<n> shrinks the persistent trace grid (blocks per CU) to leave room."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from take_amd import capi, scenes
from take_amd import cdefs as D

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W, H = 1920, 1080
sd = scenes.soup_scene(1_000_000, W, H, spp=spp, envmap=(2048, 1024))
a, b = capi.Scene(sd), capi.Scene(sd)
oa = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ob = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
half = spp // 2
# warm-up + sizes the workspaces
a.render_device(oa.data_ptr(), half, 50, seed=0, stream=sa.cuda_stream)
b.render_device(ob.data_ptr(), half, 50, seed=1, stream=sb.cuda_stream)
torch.cuda.synchronize()
t = time.perf_counter()
a.render_device(oa.data_ptr(), half, 50, seed=0, stream=sa.cuda_stream)
b.render_device(ob.data_ptr(), half, 50, seed=1, stream=sb.cuda_stream)
torch.cuda.synchronize()
serial = time.perf_counter() - t
def run(sc, out, seed, st):
    sc.render_device(out.data_ptr(), half, 50, seed=seed, stream=st.cuda_stream)
t = time.perf_counter()
ta = threading.Thread(target=run, args=(a, oa, 0, sa)); tb = threading.Thread(target=run, args=(b, ob, 1, sb))
ta.start(); tb.start(); ta.join(); tb.join()
torch.cuda.synchronize()
conc = time.perf_counter() - t
n = W * H * half * 2
print(f"TRACE_BLOCKS={os.environ.get('TAKE_HIP_TRACE_BLOCKS','-')} spp 2x{half}: serial {serial:.3f} s = {n/serial/1e6:.1f} Msamples/s | two streams {conc:.3f} s = {n/conc/1e6:.1f} Msamples/s", flush=True)
a.close(); b.close()
