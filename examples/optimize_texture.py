#!/usr/bin/env python3
"""Inverse rendering with the `prb` integrator (the reference's "gradient-based optimization" tutorial, hip_ad_rgb edition):
recover the 32 x 32 albedo texture of the Cornell box's white walls from a target image.

    python examples/optimize_texture.py [iterations]

mi.render() is a torch.autograd function: loss.backward() runs RBIntegrator.render_backward on the GPU (weight pass, primal pass,
adjoint replay) and leaves d loss / d texel in params[key].grad."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mitsuba3_amd as mi                                   # noqa: E402


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    mi.set_variant("hip_ad_rgb")
    res, tex_res, spp = 128, 32, 16
    d = mi.textured_cornell_box(res=res, tex_res=tex_res, spp=spp)
    key = "white.reflectance.data"
    # target: a checker pattern on the white walls
    i = np.arange(tex_res) * 8 // tex_res
    checker = 0.5 + 0.3 * (2 * ((i[:, None] + i[None, :]) & 1) - 1)
    d["white"]["reflectance"]["data"] = np.repeat(checker[:, :, None], 3, axis=2).astype(np.float32)
    scene = mi.load_dict(d)
    target = mi.render(scene, spp=256, seed=1000)
    params = mi.traverse(scene)
    params[key] = torch.full_like(params[key], 0.5).requires_grad_(True)            # start from a uniform grey
    params.update()
    opt = torch.optim.Adam([params[key]], lr=0.03)
    for it in range(iterations):
        opt.zero_grad()
        img = mi.render(scene, params, spp=spp, seed=it)                             # differentiable w.r.t. params[key]
        loss = ((img - target) ** 2).mean()
        loss.backward()
        opt.step()
        with torch.no_grad():
            params[key].clamp_(0.0, 1.0)
        params.update()
        err = float((params[key].detach().cpu() - torch.from_numpy(d["white"]["reflectance"]["data"])).abs().mean())
        print("iter %3d  loss %.6f  mean |texel error| %.4f" % (it, float(loss), err))


if __name__ == "__main__":
    main()
