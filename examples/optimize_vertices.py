#!/usr/bin/env python3
"""Geometry optimisation with the vertex-position gradients of `prb` (har_integrator_set_grad_positions): a tilted, lifted floor quad is
moved back to the pose that produced the target image.

    python examples/optimize_vertices.py [iterations]

Like `prb` in the reference this has no visibility-boundary term (that is `prb_reparam`): it fits smooth shading changes, here the distance
and the angle of a large floor to a small light."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3_amd as mi                                   # noqa: E402
from tests.test_shape_gradients_cpu import slab_scene         # noqa: E402  (two large slabs lit by a small light)


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    mi.set_variant("hip_ad_rgb")
    key = "floor.positions"
    d = slab_scene(mi, 64)
    d["integrator"] = {"type": "prb", "max_depth": 4, "shape_gradients": [key], "emitter_gradients": False}
    scene = mi.load_dict(d)
    target = mi.render(scene, spp=256, seed=1000)
    params = mi.traverse(scene)
    truth = params[key].clone()
    start = truth.reshape(-1, 3).clone()
    start[:, 1] += 0.3; start[1:3, 1] += 0.4                                         # lifted and tilted
    params[key] = start.clone().requires_grad_(True)                                 # N x 3, the shape of the reference's `positions` tensor
    params.update()
    opt = torch.optim.Adam([params[key]], lr=0.02)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(iterations):
        opt.zero_grad()
        img = mi.render(scene, params, spp=32, seed=it)
        loss = ((img - target) ** 2).mean()
        loss.backward()
        with torch.no_grad():                                                        # only heights move: in-plane sliding of a flat floor is a null space
            g = params[key].grad.reshape(-1, 3); g[:, 0] = 0; g[:, 2] = 0
        opt.step()
        params.update()                                                              # positions stay on the GPU: records, normals, shading triangles and the BVH refit are kernels (har_scene_update_vertices_device)
        err = (params[key].detach().reshape(-1, 3) - truth.reshape(-1, 3))[:, 1]
        # the corners are 40 units away; what the image constrains is the plane under the light: its height (mean of the corners) and slope
        print("iter %3d  loss %.6f  height error at the centre %.4f  slope error %.5f" % (it, float(loss), float(err.mean()), float((err[1:3].mean() - err[[0, 3]].mean()) / 80.0)))
    torch.cuda.synchronize()
    print("%.1f ms per optimisation step (render + backward + Adam + params.update()); accel: %s" % ((time.perf_counter() - t0) / iterations * 1e3, scene.refit_info()))


if __name__ == "__main__":
    main()
