#!/usr/bin/env python3
"""Per-kernel-class time of one PRB render_backward (primal + adjoint passes) on the bench scene (HIP events)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
textured = "textured" in sys.argv[2:]
emit = "noemit" not in sys.argv[2:]
d = mi.cornell_box() if wl == "cornell" else mi.instanced_spheres_scene(width=512, height=512, spp=256, flatten=(wl in ("flat1m", "materials1m")), materials=(wl == "materials1m"), textured=textured,
                                                                     tex_res=int(next((a[7:] for a in sys.argv[2:] if a.startswith("texres=")), 256)))
d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": emit}
scene = mi.load_dict(d); integ = scene.integrator(); integ.set_profiling(True)
g = torch.full((512, 512, 3), 1.0 / (512 * 512 * 3), device="cuda")
import time
mi.render_backward_distributed(scene, g, integ, seed=1, spp=256); torch.cuda.synchronize()
integ.set_profiling(True)
t0 = time.perf_counter()
for _ in range(3):
    mi.render_backward_distributed(scene, g, integ, seed=1, spp=256)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 3 * 1e3, "textured", textured, "emitter_gradients", emit)
print(json.dumps({k: (round(v[0], 2), v[1]) for k, v in integ.timing().items()}))
