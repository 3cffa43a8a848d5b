#!/usr/bin/env python3
"""One rank's share of the PRB step on ONE GPU: rows [k 512/N, (k + 1) 512/N) of the textured 1M-triangle scene at 512^2 x 256 spp (what a rank of an N-GPU job
differentiates before the gradient all-reduce).  usage: band_bench_prb.py [N=8] [steps=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
which = n // 2
res, spp = 512, 256
d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, textured=True)
d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True}
scene = mi.load_dict(d); integ = scene.integrator()
rows = res // n
lanes = (which * rows * res * spp, (which + 1) * rows * res * spp)
g = torch.full((res, res, 3), 1.0 / (res * res * 3), device="cuda")
wfilm = integ.render_weights(scene, 0, 1, spp)
for _ in range(2): integ.render_backward(scene, None, g, 0, 1, spp, lanes=lanes, weight_film=wfilm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): integ.render_backward(scene, None, g, 0, 1, spp, lanes=lanes, weight_film=wfilm)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("prb band %d/%d: %d lanes  %.3f ms  %.1f Mpaths/s" % (which, n, lanes[1] - lanes[0], dt * 1e3, (lanes[1] - lanes[0]) / dt / 1e6))
