#!/usr/bin/env python3
"""Recovering a height field with the vertex-position gradients of a SMOOTH-SHADED mesh: the floor is a grid with vertex normals, which a position update regenerates
(Mesh::compute_normals), so the gradient of a pixel runs through the interpolated normal of the triangle it sees and through the angle-weighted normal sums of the
whole one-ring (har_shape_grad.h: shape_item_adjoint + face_normals_adjoint).  A flat grid is bent towards the bumps that produced the target image.

    python examples/optimize_smooth_mesh.py [iterations]

No visibility-boundary term (that is `prb_reparam`): the bumps are shallow, what the image constrains is shading."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3_amd as mi                                         # noqa: E402
from tests.test_shape_gradients_cpu import smooth_slab_scene      # noqa: E402


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    mi.set_variant("hip_ad_rgb")
    key = "floor.positions"
    d = smooth_slab_scene(mi, 96, n=25)
    d["integrator"] = {"type": "prb", "max_depth": 3, "shape_gradients": [key], "emitter_gradients": False}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    truth = params[key].clone().reshape(-1, 3)
    params[key] = truth.reshape(-1).clone(); params.update()            # normals := compute_normals(positions), as after any position update
    target = mi.render(scene, spp=256, seed=1000)
    start = truth.clone(); start[:, 1] = 0.0                              # a flat floor
    params[key] = start.reshape(-1).clone().requires_grad_(True); params.update()
    inner = (truth[:, 0].abs() < 3.0) & (truth[:, 2].abs() < 3.0)        # the part of the grid the camera sees
    opt = torch.optim.Adam([params[key]], lr=0.004)
    first = None
    for it in range(iterations):
        opt.zero_grad()
        img = mi.render(scene, params, spp=32, seed=it)
        loss = ((img - target) ** 2).mean()
        loss.backward()
        with torch.no_grad():                                             # heights only, and only where the image says something
            g = params[key].grad.reshape(-1, 3); g[:, 0] = 0; g[:, 2] = 0; g[~inner] = 0
        opt.step()
        params.update()                                                   # regenerates the normals, rebuilds the acceleration structure
        first = first or float(loss.detach())
        if it % 10 == 0 or it == iterations - 1:
            # (the image constrains slopes, not heights -- shape from shading --, so the loss is what to watch: ~9x lower after 40 iterations at 32 spp)
            print("iter %3d  loss %.3e (%.1f %% of the start)" % (it, float(loss.detach()), 100 * float(loss.detach()) / first))


if __name__ == "__main__":
    main()
