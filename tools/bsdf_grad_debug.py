#!/usr/bin/env python3
"""Product vs oracle: PRB gradients of the rough models' alpha / eta / k / specular colour, per key."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import mitsuba3_amd as mi
from oracle import oracle as O
mi.set_variant("hip_ad_rgb")
res, spp = 64, 32
md = int(sys.argv[1]) if len(sys.argv) > 1 else 6
d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=12, n_v=6, flatten=True, materials=True)
if "aniso" in sys.argv:
    d["green"]["m"]["alpha_u"] = 0.12; d["green"]["m"]["alpha_v"] = 0.3; d["green"]["m"].pop("alpha", None)
if "ggxwhite" in sys.argv: d["white"]["distribution"] = "ggx"
rr = 100 if "norr" in sys.argv else 5
d["integrator"] = {"type": "prb", "max_depth": md, "rr_depth": rr, "bsdf_parameter_gradients": True}
scene = mi.load_dict(d)
osc, sensor = O.scene_from_product(scene)
grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res)
grads = scene.integrator().render_backward(scene, None, grad_in, seed=4, spp=spp)
gx, g_refl = osc.render_prb_backward_bsdf_params(sensor, grad_in, seed=4, spp=spp, max_depth=md, rr_depth=rr)
for key, (what, b) in scene._bsdf_param_keys().items():
    if what != "alpha": continue
    print(sys.argv[1:], key, what, b.kind, "product", grads[key].cpu().numpy().reshape(-1), "oracle", gx[b.index][:2].sum(), "per channel", gx[b.index][:2].tolist())
