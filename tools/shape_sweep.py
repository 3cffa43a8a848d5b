#!/usr/bin/env python3
"""Forward / PRB throughput over render shapes (resolution x spp, odd sizes, tiny jobs): looks for performance cliffs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mitsuba3_amd as mi

mi.set_variant("hip_ad_rgb")


def timed(fn, steps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps


for (w, h, spp) in [(512, 512, 256), (500, 500, 100), (33, 47, 7), (64, 64, 16), (128, 128, 16), (256, 256, 4), (1920, 1080, 16), (1000, 3, 1000), (16, 16, 4096), (4096, 4096, 1)]:
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = w; d["sensor"]["film"]["height"] = h
    scene = mi.load_dict(d)
    dt = timed(lambda: mi.render(scene, spp=spp, seed=0))
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    scene_t = mi.load_dict(mi.textured_cornell_box(res=min(w, 256), tex_res=64, spp=spp)) if w == h and w <= 256 else None
    line = {"shape": "%dx%dx%d" % (w, h, spp), "paths": w * h * spp, "fwd_ms": round(dt * 1e3, 3), "fwd_Mpaths_s": round(w * h * spp / dt / 1e6, 1)}
    if scene_t is not None:
        params = mi.traverse(scene_t); key = "white.reflectance.data"

        def step():
            p = params[key].detach().clone().requires_grad_(); params[key] = p
            img = mi.render(scene_t, params, spp=spp, seed=0); (img ** 2).mean().backward(); return p.grad
        dtp = timed(step)
        line["prb_step_ms"] = round(dtp * 1e3, 3)
    print(json.dumps(line), flush=True)
