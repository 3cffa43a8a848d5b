"""An `area` emitter whose radiance is a bitmap, on a rectangle (src/emitters/area.cpp:74, 83-90, 133-165, 185-191; emitter type 7 of include/hip_ad_rgb.h), on the
GPU against the oracle: forward images within 1e-4 with equal vertex counts, prb gradients of a reflectance bitmap lit by it within 1e-3, parameter updates,
`Integrator.sample`, and several lights at once (selection probabilities)."""
import numpy as np
import pytest

from tests.test_gpu_boundary import rel_l2
from tests.test_textured_area_light_cpu import _bitmap, lit_box

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("props", [{}, {"filter_type": "nearest", "wrap_mode": "clamp"}, {"wrap_mode": "mirror", "to_uv": "transpose"}, {"filter_type": "nearest", "to_uv": "flip"}])
def test_forward_parity_and_vertex_counts(mi, O, props):
    props = dict(props)
    if props.get("to_uv") == "transpose":
        props["to_uv"] = mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]])
    elif props.get("to_uv") == "flip":
        props["to_uv"] = mi.ScalarTransform3f([[-1, 0, 1], [0, 1, 0], [0, 0, 1]])
    scene = mi.load_dict(lit_box(mi, _bitmap(5), 48, **props))
    img = mi.render(scene, spp=32, seed=2).cpu().numpy()
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=2, spp=32, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.isfinite(img).all() and img.max() > 0
    assert rel_l2(img, ref) < 1e-4
    st = scene.integrator().stats()
    assert st["vertices"] == ost.vertices and st["paths"] == 48 * 48 * 32


def test_two_lights_one_textured_with_weights(mi, O):
    """a bitmap light next to a uniform one, unequal sampling weights: the emitter choice (scene.cpp:248-279) multiplies both strategies' densities"""
    d = lit_box(mi, _bitmap(9), 32)
    T = mi.ScalarTransform4f
    d["lamp2"] = {"type": "rectangle", "to_world": T().translate([0.4, -0.3, 0.2]).rotate([0, 1, 0], -70.0).scale([0.15, 0.2, 1.0]),
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [3.0, 6.0, 9.0]}, "sampling_weight": 0.5}}
    d["light"]["emitter"]["sampling_weight"] = 2.0
    scene = mi.load_dict(d)
    assert sorted(e["type"] for e in scene.emitters) == [0, 7]
    img = mi.render(scene, spp=32, seed=6).cpu().numpy()
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=6, spp=32, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices


def test_prb_reflectance_gradient_under_a_bitmap_light(mi, O):
    """everything the bitmap light lights is differentiable: texel gradients of a wall bitmap, image and gradient against the oracle"""
    import torch
    d = lit_box(mi, _bitmap(4), 32)
    wall = np.random.default_rng(1).uniform(0.2, 0.8, (8, 8, 3)).astype(np.float32)
    d["white"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": wall}}
    d["integrator"] = {"type": "prb", "max_depth": 5, "rr_depth": 3}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = "white.reflectance.data"
    params[key].requires_grad_()
    img = mi.render(scene, params, spp=32, seed=0)
    (img ** 2).mean().backward()
    g = params[key].grad.cpu().numpy()
    osc, sensor = O.scene_from_product(scene)
    ref, _ = osc.render_prb(sensor, seed=0, spp=32, max_depth=5, rr_depth=3)
    assert rel_l2(img.detach().cpu().numpy(), ref) < 1e-4
    grad_in = 2.0 * ref / ref.size
    _, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=mi.sample_tea_32(0, 1)[0], spp=32, max_depth=5, rr_depth=3)
    wall_index = [b for b in scene.bsdf_objs if b.id == "white"][0].tex_index
    assert np.abs(g).max() > 0 and rel_l2(g, g_tex[wall_index]) < 1e-3
    # the light's own texels are differentiable too (area.cpp:64-70; test_prb_light_texel_gradients): requires_grad switches the integrator's term on for that call
    params["light.emitter.radiance.data"].requires_grad_()
    (mi.render(scene, params, spp=4, seed=1) ** 2).mean().backward()
    assert float(params["light.emitter.radiance.data"].grad.abs().max()) > 0


@pytest.mark.parametrize("props", [{}, {"filter_type": "nearest", "wrap_mode": "clamp"}, {"wrap_mode": "mirror", "to_uv": "transpose"}, {"zeros": True}, {"two_lights": True}, {"nocache": True}])
def test_prb_light_texel_gradients(mi, O, props):
    """`radiance` of an area light is a differentiable traverse entry (area.cpp:64-70): gradients w.r.t. the texels of its bitmap (`light_texel_gradients`, har_integrator_set_grad_light_texels) --
    d Le / d radiance(si.uv) at emitter hits, d Lr_dir / d radiance(ds.uv) at visible emitter samples, committed in place by the re-shading replay -- against the oracle, texel by texel;
    the other gradients of the same call are those of the plain adjoint"""
    props = dict(props)
    if props.get("to_uv") == "transpose":
        props["to_uv"] = mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]])
    tex = _bitmap(5, 6, 5)
    if props.pop("zeros", False):          # a sample over zero texels contributes nothing and still has a derivative (its shadow ray is traced in both passes)
        tex[:3, :3] = 0.0
    two = props.pop("two_lights", False); nocache = props.pop("nocache", False)
    res = 32
    d = lit_box(mi, tex, res, **props)
    T = mi.ScalarTransform4f
    if two:                                  # a uniform light next to it: emitter choice and the colour gradient of the other light in the same call
        d["lamp2"] = {"type": "rectangle", "to_world": T().translate([0.4, -0.3, 0.2]).rotate([0, 1, 0], -70.0).scale([0.15, 0.2, 1.0]),
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [3.0, 6.0, 9.0]}, "sampling_weight": 0.5}}
    d["integrator"] = {"type": "prb", "max_depth": 5, "light_texel_gradients": True}
    if nocache:                              # the lane-indexed replay cache instead of the state tape (hide_emitters forces it): the same in-place commit
        d["integrator"]["hide_emitters"] = True
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    osc.set_hide_emitters(nocache)
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    integ = scene.integrator()
    grads = integ.render_backward(scene, None, grad_in, seed=3, spp=16)
    w_refl, w_tex, w_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=16, max_depth=5)
    key = "light.emitter.radiance.data"
    ti = [e for e in scene.emitters if e["type"] == 7][0]["light"].tex_index
    got = grads[key].cpu().numpy(); want = w_tex[ti]
    assert got.shape == tex.shape and np.abs(want).max() > 0
    assert rel_l2(got, want) < 1e-3, rel_l2(got, want)
    if not tex[:3, :3].any():
        assert np.abs(got[:2, :2]).max() > 0
    for k, (kind, b) in scene._param_keys().items():
        ref = w_emit[b] if kind == "emit" else (w_tex[b.tex_index] if kind == "tex" else w_refl[b.index])
        if np.any(ref):
            assert rel_l2(grads[k].cpu().numpy(), ref) < 1e-3, k
    # off again: the key disappears, the other gradients do not change (record tape instead of the re-shading replay)
    integ.light_texel_gradients = False
    plain = integ.render_backward(scene, None, grad_in, seed=3, spp=16)
    assert key not in plain
    for k in plain:
        assert np.allclose(plain[k].cpu().numpy(), grads[k].cpu().numpy(), rtol=2e-3, atol=1e-6 * max(1.0, float(np.abs(grads[k].cpu().numpy()).max()))), k


def test_light_texels_through_autograd(mi, O):
    """mi.render + loss.backward() with requires_grad on '<shape>.emitter.radiance.data': the integrator's switch is set for that call only; one Adam-free descent step on the texels
    lowers a least-squares loss against a target rendered with other texels"""
    import torch
    tex = _bitmap(5, 6, 5); target_tex = _bitmap(6, 6, 5)
    res = 24
    d = lit_box(mi, tex, res); d["integrator"] = {"type": "prb", "max_depth": 4}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = "light.emitter.radiance.data"
    params[key] = torch.tensor(target_tex, device="cuda"); params.update()
    target = mi.render(scene, spp=64, seed=1).detach()
    params[key] = torch.tensor(tex, device="cuda"); params.update()
    losses = []
    for step in range(2):
        p = params[key].detach().clone().requires_grad_(True); params[key] = p
        img = mi.render(scene, params, spp=64, seed=1, seed_grad=2)
        loss = ((img - target) ** 2).mean(); loss.backward()
        assert p.grad is not None and p.grad.shape == p.shape and float(p.grad.abs().max()) > 0
        losses.append(float(loss.detach()))
        with torch.no_grad():         # a small step against the gradient (2 % of the brightest texel for the steepest one)
            params[key] = (p - 0.02 * p.abs().max() * p.grad / p.grad.abs().max()).clamp(min=1e-3).detach()
        params.update()
    assert not scene.integrator().light_texel_gradients
    assert losses[1] < 0.9 * losses[0], losses          # (fixed-size normalised steps overshoot near the optimum: one step is the test)


def test_radiance_bitmap_updates_in_place(mi, O):
    import torch
    scene = mi.load_dict(lit_box(mi, _bitmap(5), 32)); mi.render(scene, spp=4, seed=0); handle = scene._h.value
    params = mi.traverse(scene)
    for tex2 in (torch.tensor(_bitmap(6)), torch.tensor(_bitmap(7), device="cuda")):
        params["light.emitter.radiance.data"] = tex2; params.update()
        assert scene._h.value == handle
        a = mi.render(scene, spp=16, seed=5).cpu().numpy()
        b = mi.render(mi.load_dict(lit_box(mi, tex2.cpu().numpy(), 32)), spp=16, seed=5).cpu().numpy()
        assert rel_l2(a, b) < 1e-6
    params["light.emitter.radiance.to_uv"] = torch.tensor(mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]]).matrix); params.update()
    assert scene._h.value == handle
    a = mi.render(scene, spp=16, seed=5).cpu().numpy()
    b = mi.render(mi.load_dict(lit_box(mi, _bitmap(7), 32, to_uv=mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]]))), spp=16, seed=5).cpu().numpy()
    assert rel_l2(a, b) < 1e-6
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=5, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(a, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices


def test_c_abi_setters_refuse_before_they_change_anything(mi, O):
    """har_scene_set_texture / _set_texture_device / _set_texture_to_uv on a bitmap that an area light radiates: texels without luminance, negative texels and a to_uv
    that does not keep the unit square are refused by the library itself, and the scene still renders what it rendered before"""
    import torch
    from mitsuba3_amd import _capi
    from mitsuba3_amd.core import lib, _fp, _f32, _ptr
    scene = mi.load_dict(lit_box(mi, _bitmap(5), 24)); before = mi.render(scene, spp=8, seed=1).cpu().numpy()
    idx = scene.emitters[0]["radiance_texture"]
    zero = np.zeros_like(scene.textures[idx]); neg = -np.ones_like(zero)
    assert lib().har_scene_set_texture(scene._h, idx, _fp(zero)) != 0 and b"luminance" in lib().har_last_error()
    assert lib().har_scene_set_texture(scene._h, idx, _fp(neg)) != 0 and b"non-negative" in lib().har_last_error()
    z = torch.zeros(zero.shape, device="cuda")
    assert lib().har_scene_set_texture_device(scene._h, idx, _ptr(z), None) != 0
    assert lib().har_scene_set_texture_to_uv(scene._h, idx, _fp(_f32([2, 0, 0, 0, 1, 0]))) != 0 and b"unit square" in lib().har_last_error()
    assert np.array_equal(mi.render(scene, spp=8, seed=1).cpu().numpy(), before) or rel_l2(mi.render(scene, spp=8, seed=1).cpu().numpy(), before) < 1e-6
