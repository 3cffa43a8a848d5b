#!/usr/bin/env python3
"""One rank's share of the headline frame on ONE GPU: rows [0, 512/N) of the 1M-triangle scene at 512^2 x 256 spp (what a rank of an N-GPU job
renders before the film reduce).  usage: band_bench.py [N=8] [frames=10]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = int(sys.argv[3]) if len(sys.argv) > 3 else n // 2        # a band from the middle of the frame (the expensive rows)
res, spp = 512, 256
scene = mi.load_dict(mi.instanced_spheres_scene(width=res, height=res, spp=spp))
integ = scene.integrator()
rows = res // n
lanes = (which * rows * res * spp, (which + 1) * rows * res * spp)
film = integ.render_film(scene, seed=0, spp=spp, lanes=lanes); torch.cuda.synchronize()
for _ in range(2): integ.render_film(scene, seed=0, spp=spp, lanes=lanes)
torch.cuda.synchronize()
integ.set_profiling(True)
t0 = time.perf_counter()
for _ in range(frames): integ.render_film(scene, seed=0, spp=spp, lanes=lanes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / frames
t = integ.timing()
print("band %d/%d: %d lanes  %.3f ms  %.1f Mpaths/s   kernels %s" % (which, n, lanes[1] - lanes[0], dt * 1e3, (lanes[1] - lanes[0]) / dt / 1e6,
      json.dumps({k: round(v[0], 3) for k, v in t.items()})))
