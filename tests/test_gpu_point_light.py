"""`point`, `spot` and `directional` emitters (src/emitters/point.cpp, spot.cpp, directional.cpp; HarEmitter types 4 / 5 / 6) on the device, through the C ABI, against the oracle:
forward images (1e-4) with equal path / vertex / ray counters, prb gradients (1e-3) w.r.t. albedos, a bitmap albedo and the light's intensity,
har_integrator_sample, and src/render/tests/test_ad.py:55-134 literally through mi.render + autograd (backward and forward mode)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def lit_box(mi, res, textured=False, materials=False, only_points=False):
    """the Cornell box with two point lights next to (or instead of) its area light: three emitters exercise the uniform emitter choice with sample re-use
    (scene.cpp:248-271); `materials`: the instanced spheres with rough / dielectric BSDFs around the bulbs (generic shading kernel)"""
    if materials:
        d = mi.instanced_spheres_scene(width=res, height=res, spp=4, grid=3, n_u=16, n_v=8, materials=True)
    else:
        d = mi.textured_cornell_box(res=res, tex_res=16, spp=4) if textured else mi.cornell_box()
        d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    if only_points:
        for k in [k for k, v in d.items() if isinstance(v, dict) and isinstance(v.get("emitter"), dict)]:
            del d[k]["emitter"]
    d["bulb"] = {"type": "point", "position": [0.3, 0.2, 0.1], "intensity": {"type": "rgb", "value": [0.5, 0.4, 0.3]}}
    d["bulb2"] = {"type": "point", "to_world": mi.ScalarTransform4f().translate([-0.5, -0.4, 0.5]), "intensity": 0.2}
    return d


@pytest.mark.parametrize("kind", ["diffuse", "materials", "only_points"])
def test_forward_parity_with_point_lights(mi, O, kind):
    res, spp = 64, 16
    scene = mi.load_dict(lit_box(mi, res, materials=kind == "materials", only_points=kind == "only_points"))
    assert [e.get("type", 0) for e in scene.emitters].count(4) == 2
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=5).cpu().numpy()
    st = scene.integrator().stats()
    ref, ost = osc.render_path(sensor, seed=5, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.abs(ref).max() > 0 and rel_l2(img, ref) < 1e-4, (kind, rel_l2(img, ref))          # north_star forward tolerance
    assert st["paths"] == res * res * spp and st["vertices"] == ost.vertices, (kind, st, ost.vertices)
    # the oracle sends a shadow ray for every emitter sample with pdf != 0 (always, for a point light); the device skips the ones whose contribution is exactly zero
    assert 0 < st["shadow_rays"] <= ost.shadow_rays, (kind, st, ost.shadow_rays)


def test_prb_gradients_with_point_lights(mi, O):
    """albedo texture, constant albedos, the area light's radiance AND the point lights' intensities (PointLight::traverse: `intensity` is differentiable)"""
    res, spp, md = 48, 16, 6
    d = lit_box(mi, res, textured=True); d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    keys = scene._param_keys()
    assert "bulb.intensity.value" in keys and "bulb2.intensity.value" in keys
    grad_in = np.random.default_rng(7).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
    assert len(ek) == 3
    got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([g_emit[i] for i in ek.values()])
    assert np.abs(want).min() > 0 and rel_l2(got, want) < 1e-3, (got, want)                        # north_star PRB tolerance
    for k, (kind, b) in keys.items():
        if kind == "tex":
            assert rel_l2(grads[k].cpu().numpy(), g_tex[b.tex_index]) < 1e-3, k
        elif kind == "rgb":
            assert rel_l2(grads[k].cpu().numpy(), g_refl[b.index]) < 1e-3, k
    # forward mode: the tangent image of one intensity equals the finite difference of the primal (the image is linear in it)
    import torch
    integ = scene.integrator()
    fwd = integ.render_forward(scene, seed=3, spp=spp, tangents={"bulb.intensity.value": torch.ones(3, device="cuda")}).cpu().numpy()
    params = mi.traverse(scene)
    a = mi.render(scene, spp=spp, seed=3).cpu().numpy().astype(np.float64)
    params["bulb.intensity.value"] = params["bulb.intensity.value"] + 1.0; params.update()
    b = mi.render(scene, spp=spp, seed=3).cpu().numpy().astype(np.float64)
    assert rel_l2(fwd, b - a) < 2e-3


def test_integrator_sample_with_a_point_light(mi, O):
    """Integrator::sample through the C ABI (har_integrator_sample) on rays into a scene lit by point lights only, path and prb"""
    n = 20000
    rng = np.random.default_rng(1)
    o = np.tile(np.array([[0.0], [0.0], [3.9]], np.float32), (1, n)); d = rng.normal(size=(3, n)).astype(np.float32); d[2] = -np.abs(d[2]) - 1.0
    d /= np.linalg.norm(d, axis=0); d = np.ascontiguousarray(d, np.float32); maxt = np.full(n, 3.402823466e+38, np.float32)
    for kind in ("path", "prb"):
        dd = lit_box(mi, 16, only_points=True); dd["integrator"] = {"type": kind, "max_depth": 5, "rr_depth": 3}
        scene = mi.load_dict(dd)
        osc, _ = O.scene_from_product(scene)
        sampler = mi.Sampler({"sample_count": 4, "seed": 5}); sampler.seed(3, n)
        spec, valid = scene.integrator().sample(scene, sampler, mi.Ray3f(o, d, maxt))
        ref, rvalid, _ = osc.integrator_sample(o, d, maxt, seed=5 + 3, max_depth=5, rr_depth=3, prb=(kind == "prb"))
        assert np.array_equal(valid.cpu().numpy().astype(np.uint8), rvalid), kind
        assert np.abs(ref).max() > 0 and rel_l2(spec.cpu().numpy(), ref) < 1e-4, kind


def spot_box(mi, res, textured=False):
    d = mi.textured_cornell_box(res=res, tex_res=16, spp=4) if textured else mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["spot"] = {"type": "spot", "cutoff_angle": 40.0, "beam_width": 25.0, "intensity": {"type": "rgb", "value": [3.0, 2.0, 1.0]},
                 "to_world": mi.ScalarTransform4f().look_at(origin=[0.3, 0.9, 0.2], target=[-0.2, -1.0, 0.1], up=[0, 0, 1])}
    return d


def test_forward_parity_with_a_spot_light(mi, O):
    """SpotLight::sample_direction (spot.cpp:177-211): falloff between beam width and cut-off, the inverse transform, MIS weight 1"""
    res, spp = 64, 16
    scene = mi.load_dict(spot_box(mi, res))
    assert [e.get("type", 0) for e in scene.emitters].count(5) == 1
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=5).cpu().numpy()
    st = scene.integrator().stats()
    ref, ost = osc.render_path(sensor, seed=5, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.abs(ref).max() > 0 and rel_l2(img, ref) < 1e-4, rel_l2(img, ref)
    assert st["paths"] == res * res * spp and st["vertices"] == ost.vertices
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = res; d2["sensor"]["film"]["height"] = res
    plain = mi.render(mi.load_dict(d2), spp=spp, seed=5).cpu().numpy()
    assert rel_l2(plain, ref) > 0.02                              # the spot light is in the picture


def test_prb_gradients_with_a_spot_light(mi, O):
    res, spp, md = 48, 16, 6
    d = spot_box(mi, res, textured=True); d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    keys = scene._param_keys()
    assert "spot.intensity.value" in keys
    grad_in = np.random.default_rng(8).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
    got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([g_emit[i] for i in ek.values()])
    assert len(ek) == 2 and np.abs(want).min() > 0 and rel_l2(got, want) < 1e-3, (got, want)
    for k, (kind, b) in keys.items():
        if kind == "tex":
            assert rel_l2(grads[k].cpu().numpy(), g_tex[b.tex_index]) < 1e-3, k
        elif kind == "rgb":
            assert rel_l2(grads[k].cpu().numpy(), g_refl[b.index]) < 1e-3, k


def sun_box(mi, res, textured=False):
    d = mi.textured_cornell_box(res=res, tex_res=16, spp=4) if textured else mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["sun"] = {"type": "directional", "direction": [0.3, -0.2, -1.0], "irradiance": {"type": "rgb", "value": [2.0, 1.5, 1.0]}}      # through the open front of the box
    return d


def test_forward_parity_with_a_directional_light(mi, O):
    """DirectionalEmitter::sample_direction (directional.cpp:149-176): shadow rays of twice the bounding sphere's radius, MIS weight 1, no attenuation"""
    res, spp = 64, 16
    scene = mi.load_dict(sun_box(mi, res))
    assert [e.get("type", 0) for e in scene.emitters].count(6) == 1
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=5).cpu().numpy()
    st = scene.integrator().stats()
    ref, ost = osc.render_path(sensor, seed=5, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.abs(ref).max() > 0 and rel_l2(img, ref) < 1e-4, rel_l2(img, ref)
    assert st["paths"] == res * res * spp and st["vertices"] == ost.vertices
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = res; d2["sensor"]["film"]["height"] = res
    plain = mi.render(mi.load_dict(d2), spp=spp, seed=5).cpu().numpy()
    assert rel_l2(plain, ref) > 0.02


def test_prb_gradients_with_a_directional_light(mi, O):
    res, spp, md = 48, 16, 6
    d = sun_box(mi, res, textured=True); d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    keys = scene._param_keys()
    assert "sun.irradiance.value" in keys
    grad_in = np.random.default_rng(9).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
    got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([g_emit[i] for i in ek.values()])
    assert len(ek) == 2 and np.abs(want).min() > 0 and rel_l2(got, want) < 1e-3, (got, want)
    for k, (kind, b) in keys.items():
        if kind == "tex":
            assert rel_l2(grads[k].cpu().numpy(), g_tex[b.tex_index]) < 1e-3, k
        elif kind == "rgb":
            assert rel_l2(grads[k].cpu().numpy(), g_refl[b.index]) < 1e-3, k


def simple_scene(mi, res=1, integrator="prb"):
    """make_simple_scene of src/render/tests/test_ad.py:6-43 (the AD-capable integrator of this variant is `prb`)"""
    return {'type': 'scene', "integrator": {"type": integrator},
            "mysensor": {"type": "perspective", "near_clip": 0.1, "far_clip": 1000.0,
                         "to_world": mi.ScalarTransform4f().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                         "myfilm": {"type": "hdrfilm", "rfilter": {"type": "box"}, "width": res, "height": res},
                         "mysampler": {"type": "independent", "sample_count": 1}},
            'rect': {'type': 'rectangle', "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.6, 0.6, 0.6]}}},
            "emitter": {"type": "point", "position": [0, 0, 5]}}


@pytest.mark.parametrize("spp", [1, 4, 44])
def test01_bsdf_reflectance_backward(mi, spp):
    """src/render/tests/test_ad.py:55-95: one step of gradient descent on a linear function.  The reference differentiates
    `integrator.render(scene, seed=0, spp)` itself, i.e. primal and adjoint share their samples: render_backward with the primal's seed (mi.render's autograd node
    re-seeds the adjoint pass, util.py:505-507, and would give an independent estimate of the same gradient)"""
    import torch
    scene = mi.load_dict(simple_scene(mi))
    key = 'rect.bsdf.reflectance.value'
    params = mi.traverse(scene)
    img_1 = mi.render(scene, seed=0, spp=spp)
    loss = img_1.sum()
    grad = scene.integrator().render_backward(scene, None, np.ones((1, 1, 3), np.float32), seed=0, spp=spp)[key]      # d sum(img) / d reflectance
    lr = 0.01
    v = params[key].detach().clone(); v[0] += lr
    params[key] = v; params.update()
    img_2 = mi.render(scene, seed=0, spp=spp)
    new_loss = img_2.sum()
    assert torch.allclose(loss, new_loss - lr * grad[0], rtol=1e-5, atol=1e-8), (float(loss), float(new_loss), float(grad[0]))
    assert float(loss) > 0 or spp == 1
    # and through mi.render + autograd (an independent estimate: same expectation, its own samples)
    if spp == 44:
        params[key] = params[key].detach().clone().requires_grad_()
        img = mi.render(scene, params, seed=0, spp=4096)
        img.sum().backward()
        assert abs(float(params[key].grad[0]) / (float(img.detach().sum()) / 3 / float(params[key].detach()[0])) - 1) < 0.05


@pytest.mark.parametrize("spp", [1, 4])
def test02_bsdf_reflectance_forward(mi, spp):
    """src/render/tests/test_ad.py:98-134: the forward-mode derivative image of a reflectance shifted by X predicts the image at X + lr"""
    import torch
    scene = mi.load_dict(simple_scene(mi))
    key = 'rect.bsdf.reflectance.value'
    params = mi.traverse(scene)
    params[key] = params[key] + 0.1; params.update()
    img_1 = mi.render(scene, seed=0, spp=spp)
    grad = scene.integrator().render_forward(scene, seed=0, spp=spp, tangents={key: torch.ones(3, device="cuda")})
    lr = 0.1
    params[key] = params[key] + lr; params.update()
    img_2 = mi.render(scene, seed=0, spp=spp)
    assert torch.allclose(img_1, img_2 - lr * grad, rtol=1e-5, atol=1e-8)
