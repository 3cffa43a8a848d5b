#!/usr/bin/env python3
"""Times the BASELINE.json configurations that are not the bench line (configs 2, 3, 4) on one GPU and prints one JSON
line each (kept under profiles/).  Synthetic scenes of SURVEY.md 8(d); scene + BVH build excluded, develop included."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                            # noqa: E402
import mitsuba3_amd as mi                               # noqa: E402


def timed(fn, steps=2, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def c5():
    """config 5 on ONE GPU: Cornell 4096x4096x1024 spp = 2^34 paths as 8 passes of 128 spp (with N ranks each renders
    1/N of the per-pass wavefront and the films are summed by one RCCL reduce).  16 GiB of sampler state lives across the passes."""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 4096; d["sensor"]["film"]["height"] = 4096
    d["integrator"] = {"type": "path", "max_depth": 8, "samples_per_pass": 128}
    scene = mi.load_dict(d)
    mi.render(scene, spp=128, seed=0)                 # warm-up: one pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = mi.render(scene, spp=1024, seed=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = scene.integrator().stats()
    print(json.dumps({"config": "C5 cornell path 4096x4096x1024spp, 8 passes of 128 spp, 1 GPU", "seconds": dt, "Mpaths_per_s": 2.0 ** 34 / dt / 1e6,
                      "paths": int(st["paths"]), "image_mean": float(img.mean()), "finite": bool(torch.isfinite(img).all()),
                      "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30}))


def main():
    mi.set_variant("hip_ad_rgb")
    if "--c5" in sys.argv:
        return c5()
    out = []
    # config 2: Cornell box, path, 512x512x256 spp
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
    scene = mi.load_dict(d)
    dt = timed(lambda: mi.render(scene, spp=256, seed=0))
    out.append({"config": "C2 cornell path 512x512x256spp", "seconds": dt, "Mpaths_per_s": 512 * 512 * 256 / dt / 1e6})
    # config 3: 1M-triangle instanced scene, path, 1024x1024x512 spp (32 chunks of 2^24 lanes)
    scene = mi.load_dict(mi.instanced_spheres_scene(width=1024, height=1024, spp=512))
    dt = timed(lambda: mi.render(scene, spp=512, seed=0), steps=1, warmup=1)
    st = scene.integrator().stats()
    out.append({"config": "C3 instanced 1M-tri path 1024x1024x512spp", "seconds": dt, "Mpaths_per_s": 1024 * 1024 * 512 / dt / 1e6,
                "paths": int(st["paths"]), "accel": scene.accel_info()})
    # config 4: PRB inverse render step: Cornell box, 256x256 albedo texture, 256x256 film, 256 spp, loss = mean(img^2)
    scene = mi.load_dict(mi.textured_cornell_box(res=256, tex_res=256, spp=256))
    params = mi.traverse(scene); key = "white.reflectance.data"

    def step():
        p = params[key].detach().clone().requires_grad_()
        params[key] = p
        img = mi.render(scene, params, spp=256, seed=0)
        (img ** 2).mean().backward()
        return p.grad
    dt = timed(step)
    g = step()
    out.append({"config": "C4 prb texture gradient 256x256x256spp (primal render + backward)", "seconds": dt,
                "Mpaths_per_s_per_pass": 256 * 256 * 256 / dt / 1e6, "grad_abs_max": float(g.abs().max()), "grad_finite": bool(torch.isfinite(g).all())})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
