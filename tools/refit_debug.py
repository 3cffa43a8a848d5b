import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
d = mi.instanced_spheres_scene(width=48, height=48, spp=16, grid=4, n_u=40, n_v=20, flatten=False)
d.pop("ceiling"); d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.4, 0.5, 0.6]}}
scene = mi.load_dict(d); mi.render(scene, spp=4, seed=0)
params = mi.traverse(scene)
key = "spheres.ball.positions"
rng = np.random.default_rng(5)
for amount in (0.002, 0.01, 0.04):
    new = params[key].cpu().numpy() + rng.normal(scale=amount, size=params[key].shape).astype(np.float32)
    params[key] = torch.tensor(new, device="cuda"); params.update()
    print("handle", scene._h, scene.refit_info() if scene._h is not None else None)
