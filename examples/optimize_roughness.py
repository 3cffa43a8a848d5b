#!/usr/bin/env python3
"""Material recovery with the `prb` integrator: the roughness of the rough-plastic walls and the complex index of refraction of the conductor
spheres of a small scene, from a target image.

    python examples/optimize_roughness.py [iterations]

requires_grad on 'white.alpha' / 'green.brdf_0.eta.value' is all it takes: mi.render() switches the corresponding adjoint terms on
(hand-derived d/d alpha of the microfacet distribution and shadowing terms, d/d eta of the conductor Fresnel term)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mitsuba3_amd as mi                                   # noqa: E402


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    mi.set_variant("hip_ad_rgb")
    res, spp = 96, 32
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=24, n_v=12, flatten=True, materials=True)
    d["integrator"] = {"type": "prb", "max_depth": 5, "rr_depth": 5}
    scene = mi.load_dict(d)
    target = mi.render(scene, spp=512, seed=1000).detach()
    params = mi.traverse(scene)
    truth = {k: params[k].clone() for k in ("white.alpha", "green.brdf_0.eta.value")}
    params["white.alpha"] = torch.full_like(truth["white.alpha"], 0.45).requires_grad_(True)
    params["green.brdf_0.eta.value"] = (truth["green.brdf_0.eta.value"] * 0.0 + 0.6).requires_grad_(True)
    params.update()
    opt = torch.optim.Adam([{"params": [params["white.alpha"]], "lr": 0.01}, {"params": [params["green.brdf_0.eta.value"]], "lr": 0.02}])
    for it in range(iterations):
        opt.zero_grad()
        img = mi.render(scene, params, spp=spp, seed=it)
        loss = ((img - target) ** 2).mean()
        loss.backward()
        opt.step()
        with torch.no_grad():
            params["white.alpha"].clamp_(0.02, 1.0); params["green.brdf_0.eta.value"].clamp_(0.05, 5.0)
        params.update()
        print("iter %3d  loss %.6f  alpha %.4f (true %.4f)  eta %s (true %s)" % (
            it, float(loss), float(params["white.alpha"].detach()[0]), float(truth["white.alpha"][0]),
            [round(float(x), 3) for x in params["green.brdf_0.eta.value"].detach().cpu()], [round(float(x), 3) for x in truth["green.brdf_0.eta.value"].cpu()]))


if __name__ == "__main__":
    main()
