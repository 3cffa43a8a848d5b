import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mitsuba3_amd as mi
from mitsuba3_amd.scenes import bumpy_sphere
mi.set_variant("hip_ad_rgb")
def loop(host):
    if host: os.environ["HAR_HOST_VERTEX_UPDATE"] = "1"
    else: os.environ.pop("HAR_HOST_VERTEX_UPDATE", None)
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["integrator"] = {"type": "prb", "max_depth": 4}
    P, N, UV, F = bumpy_sphere(n_u=32, n_v=16, radius=0.35)
    d.pop("small-box"); d.pop("large-box")
    d["blob"] = {"type": "mesh", "positions": P + np.array([0.0, -0.45, 0.0], np.float32), "normals": N, "faces": F, "bsdf": {"type": "ref", "id": "white"}}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = "blob.vertex_positions"
    params[key] = params[key].clone().requires_grad_(True); params.update()
    opt = torch.optim.SGD([params[key]], lr=1e-3)
    for it in range(3):
        opt.zero_grad()
        img = mi.render(scene, params, spp=16, seed=it)
        (img ** 2).mean().backward()
        g = params[key].grad
        print("host" if host else "dev ", it, "img mean %.6f" % float(img.mean()), "grad absmax %.4g" % float(g.abs().max()), "finite", bool(torch.isfinite(g).all()), "handle", scene._h is not None, "rebuilds", getattr(scene, "accel_rebuilds", 0))
        opt.step(); params.update()
loop(False); loop(True)
