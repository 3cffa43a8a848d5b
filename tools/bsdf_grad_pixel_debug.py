#!/usr/bin/env python3
"""Per-pixel (one sample) comparison of the alpha gradient of the Beckmann rough plastic, product vs oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mitsuba3_amd as mi
from oracle import oracle as O
mi.set_variant("hip_ad_rgb")
res, spp, md = 64, 32, 3
d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=12, n_v=6, flatten=True, materials=True)
d["integrator"] = {"type": "prb", "max_depth": md, "rr_depth": 100, "bsdf_parameter_gradients": True}
scene = mi.load_dict(d)
osc, sensor = O.scene_from_product(scene)
integ = scene.integrator()
(key, (what, b)), = [(k, v) for k, v in scene._bsdf_param_keys().items() if k == "white.alpha"]
def both(g):
    got = float(integ.render_backward(scene, None, g, seed=4, spp=spp)[key].cpu().numpy()[0])
    gx, _ = osc.render_prb_backward_bsdf_params(sensor, g, seed=4, spp=spp, max_depth=md, rr_depth=100)
    return got, float(gx[b.index][:2].sum())
rows = []
for y in range(res):
    g = np.zeros((res, res, 3), np.float32); g[y] = 1.0
    got, ref = both(g)
    if abs(got - ref) > 1e-3 * abs(ref) + 1e-7: rows.append(y)
print("bad rows", rows)
bad = []
for y in rows[:4]:
    for x in range(res):
        g = np.zeros((res, res, 3), np.float32); g[y, x] = 1.0
        got, ref = both(g)
        if abs(got - ref) > 1e-3 * abs(ref) + 1e-7: bad.append((y * res + x, got, ref))
print("mismatching pixels", len(bad))
for p, got, ref in bad[:6]:
    print("PIXEL", p, "product", got, "oracle", ref, flush=True)
    g = np.zeros((res, res, 3), np.float32); g[p // res, p % res] = 1.0
    os.environ["ORC_DEBUG_EXTRA"] = "1"
    osc.render_prb_backward_bsdf_params(sensor, g, seed=4, spp=spp, max_depth=md, rr_depth=100, threads=1)
    del os.environ["ORC_DEBUG_EXTRA"]
    sys.stderr.flush()
