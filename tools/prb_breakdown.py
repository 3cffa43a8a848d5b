#!/usr/bin/env python3
"""Per-kernel-class time of one PRB render_backward (primal + adjoint passes) on the bench scene (HIP events)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
d = mi.cornell_box() if wl == "cornell" else mi.instanced_spheres_scene(width=512, height=512, spp=256, flatten=(wl in ("flat1m", "materials1m")), materials=(wl == "materials1m"))
d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5}
scene = mi.load_dict(d); integ = scene.integrator(); integ.set_profiling(True)
g = torch.full((512, 512, 3), 1.0 / (512 * 512 * 3), device="cuda")
for _ in range(2):
    mi.render_backward_distributed(scene, g, integ, seed=1, spp=256)
torch.cuda.synchronize()
print(json.dumps({k: (round(v[0], 2), v[1]) for k, v in integ.timing().items()}))
