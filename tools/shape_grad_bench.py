#!/usr/bin/env python3
"""PRB step with and without vertex-position gradients (Cornell box of flat meshes, 256^2 x 64 spp; and the smooth bumpy floor): what the shape adjoint costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
from tests.test_shape_gradients_cpu import cbox_mesh_scene, smooth_slab_scene
for name, d, keys in (("cbox_mesh", cbox_mesh_scene(mi, 256), ["small-box", "large-box", "floor"]), ("smooth_floor", smooth_slab_scene(mi, 256, n=65), ["floor"])):
    d["sensor"]["sampler"]["sample_count"] = 64
    for on in (False, True):
        d["integrator"] = {"type": "prb", "max_depth": 6, "shape_gradients": [k + ".positions" for k in keys] if on else False}
        scene = mi.load_dict(d); integ = scene.integrator()
        g = torch.full((256, 256, 3), 1.0 / (256 * 256 * 3), device="cuda")
        for _ in range(2): integ.render_backward(scene, None, g, seed=1, spp=64)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): integ.render_backward(scene, None, g, seed=1, spp=64)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-14s shape_gradients=%-5s %.2f ms  %.1f Mpaths/s" % (name, on, dt * 1e3, 256 * 256 * 64 / dt / 1e6))
