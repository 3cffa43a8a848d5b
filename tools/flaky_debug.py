"""flaky-failure hunt: the two test_forward_parity_1m_scene_512 cases back to back in one process (1 M lanes = two-stream mode)"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitsuba3_amd as mi
from oracle import oracle as O
mi.set_variant("hip_ad_rgb")
res, spp = 512, 4
use_oracle = "oracle" in sys.argv
bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for flatten in (False, True):
        scene = mi.load_dict(mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=flatten))
        if use_oracle:
            osc, sensor = O.scene_from_product(scene)
        img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
        st1 = scene.integrator().stats()
        if use_oracle:
            ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
        st2 = scene.integrator().stats()
        if st1["paths"] != res * res * spp or st2 != st1:
            bad += 1; print("iter", k, "flatten", flatten, "st1", st1, "st2", st2, flush=True)
print("bad:", bad)
