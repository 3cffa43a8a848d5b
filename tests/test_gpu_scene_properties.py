"""Round-5 rows a10 / a13 on the device, through the C ABI, against the oracle:
 * emitters with `sampling_weight` (Scene::m_emitter_distr, src/render/scene.cpp:120-141, 248-279, 378-388): forward 1e-4 with equal vertex counts, prb
   gradients 1e-3 (albedos, bitmap texels, emitter radiances), weights that include an environment emitter and a never-sampled emitter;
 * bitmap `to_uv` (src/textures/bitmap.cpp:175, 565, 847): forward, texel gradients through the transformed taps, vertex-position gradients through the
   transformed lookup (bitmap.cpp:591-598)."""
import numpy as np
import pytest

from tests.test_scene_properties_cpu import two_light_scene, textured_scene

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.mark.parametrize("weights", [(1.0, 3.0, None), (0.25, 1.0, None), (2.0, 2.0, None), (1.0, 3.0, 0.5), (1.0, 0.0, None)])
def test_forward_parity_with_weighted_emitters(mi, O, weights):
    res, spp = 64, 32
    scene = mi.load_dict(two_light_scene(mi, weights[0], weights[1], res=res, sky=weights[2]))
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=5).cpu().numpy()
    st = scene.integrator().stats()
    ref, ost = osc.render_path(sensor, seed=5, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.abs(ref).max() > 0 and rel_l2(img, ref) < 1e-4, (weights, rel_l2(img, ref))            # north_star forward tolerance
    assert st["paths"] == res * res * spp and st["vertices"] == ost.vertices, (weights, st, ost.vertices)
    if weights[:2] != (2.0, 2.0):
        # ... and it is not the uniform-selection picture: ignoring the property would be seen
        uni = mi.render(mi.load_dict(two_light_scene(mi, 1.0, 1.0, res=res, sky=None if weights[2] is None else 1.0)), spp=spp, seed=5).cpu().numpy()
        assert rel_l2(uni, ref) > 1e-3


def test_single_weighted_emitter_and_materials(mi, O):
    """one emitter of weight 3 (pdf_emitter = 3 * (1 / 3), scene.cpp:326-338) on the materials scene: the generic shading kernel with a distribution"""
    res, spp = 48, 16
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=16, n_v=8, materials=True)
    lights = [k for k, v in d.items() if isinstance(v, dict) and isinstance(v.get("emitter"), dict)]
    assert lights
    for k in lights:
        d[k]["emitter"]["sampling_weight"] = 3.0
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=2).cpu().numpy()
    ref, ost = osc.render_path(sensor, seed=2, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices


def test_prb_gradients_with_weighted_emitters(mi, O):
    res, spp, md = 48, 32, 6
    d = two_light_scene(mi, 1.0, 3.0, res=res, sky=0.5)
    tex = np.random.default_rng(2).uniform(0.2, 0.9, (16, 16, 3)).astype(np.float32)
    d["white"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex}}
    d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    keys = scene._param_keys()
    grad_in = np.random.default_rng(7).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
    assert len(ek) == 3
    got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([g_emit[i] for i in ek.values()])
    assert np.abs(want).min() > 0 and rel_l2(got, want) < 1e-3, (got, want)                          # north_star PRB tolerance
    n_tex = 0
    for k, (kind, b) in keys.items():
        if kind == "tex":
            assert rel_l2(grads[k].cpu().numpy(), g_tex[b.tex_index]) < 1e-3, k; n_tex += 1
        elif kind == "rgb":
            assert rel_l2(grads[k].cpu().numpy(), g_refl[b.index]) < 1e-3, k
    assert n_tex == 1
    # the primal image of the prb integrator
    img = scene.integrator().render(scene, seed=3, spp=spp).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=3, spp=spp, max_depth=md)
    assert rel_l2(img, ref) < 1e-4


@pytest.mark.parametrize("filter_type,wrap_mode", [("bilinear", "repeat"), ("nearest", "mirror"), ("bilinear", "clamp")])
def test_to_uv_forward_and_texel_gradients(mi, O, filter_type, wrap_mode):
    res, spp, md = 48, 32, 6
    T = mi.ScalarTransform3f
    d, tex = textured_scene(mi, T().translate([0.1, 0.3]).rotate(-20.0).scale([3.0, 2.0]), res=res, filter_type=filter_type, wrap_mode=wrap_mode, tex_res=16)
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=9).cpu().numpy()
    ref, ost = osc.render_path(sensor, seed=9, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices
    d0, _ = textured_scene(mi, None, res=res, filter_type=filter_type, wrap_mode=wrap_mode, tex_res=16)
    plain = mi.render(mi.load_dict(d0), spp=spp, seed=9).cpu().numpy()
    assert rel_l2(plain, ref) > 1e-2                                     # the transform is in the picture
    d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    k = [k for k, v in scene._param_keys().items() if v[0] == "tex"][0]
    ti = scene._param_keys()[k][1].tex_index
    assert np.abs(g_tex[ti]).max() > 0 and rel_l2(grads[k].cpu().numpy(), g_tex[ti]) < 1e-3


def test_to_uv_through_mi_render_autograd_and_update(mi, O):
    """the public route: mi.traverse -> requires_grad -> mi.render -> backward, then params.update() with new texels: the updated scene renders like a fresh one"""
    import torch
    T = mi.ScalarTransform3f
    d, tex = textured_scene(mi, T().rotate(40.0).scale([2.0, 2.0]), res=32, tex_res=8)
    d["integrator"] = {"type": "prb", "max_depth": 5}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = [k for k in params if k.endswith(".data")][0]
    params[key].requires_grad_()
    out = mi.render(scene, params, spp=16, seed=1)
    (out ** 2).mean().backward()
    g = params[key].grad.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    new = (tex * 0.5 + 0.25).astype(np.float32)
    params[key] = torch.tensor(new, device="cuda"); params.update()
    a = mi.render(scene, spp=16, seed=1).detach().cpu().numpy()
    d2, _ = textured_scene(mi, T().rotate(40.0).scale([2.0, 2.0]), res=32, tex_res=8)
    d2["white"]["reflectance"]["data"] = new; d2["integrator"] = {"type": "prb", "max_depth": 5}
    b = mi.render(mi.load_dict(d2), spp=16, seed=1).cpu().numpy()
    assert rel_l2(a, b) < 1e-6                 # the film is summed with float atomics: equal up to the order of the additions


def test_vertex_position_gradients_through_a_transformed_texture(mi, O):
    """shape gradients of a mesh whose albedo is a bitmap with `to_uv`: d rho / d uv runs through the transpose of the transform's linear part (bitmap.cpp:591-598)"""
    res, spp, md = 32, 64, 4
    T = mi.ScalarTransform3f
    d, tex = textured_scene(mi, T().rotate(30.0).scale([1.5, 2.5]), res=res, tex_res=8)
    d["integrator"] = {"type": "prb", "max_depth": md, "shape_gradients": ["floor.positions"]}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(8).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    mesh = scene._position_keys()["floor.positions"]
    want, _, w_tex, _ = osc.render_prb_backward_shape(sensor, grad_in, [mesh], seed=3, spp=spp, max_depth=md)
    got = grads["floor.positions"].cpu().numpy().reshape(-1, 3)
    scale = np.abs(want[mesh]).max()
    assert scale > 0 and np.abs(got - want[mesh]).max() < 2e-3 * scale, np.abs(got - want[mesh]).max() / scale
    k = [k for k, v in scene._param_keys().items() if v[0] == "tex"][0]
    assert rel_l2(grads[k].cpu().numpy(), w_tex[scene._param_keys()[k][1].tex_index]) < 1e-3


@pytest.mark.parametrize("weights", [[1.0, 1.0, 1.0], [1.3, 3.8, 0.0]])
def test_scene_sample_emitter_and_pdf_emitter_on_device(mi, O, weights):
    """Scene::sample_emitter / pdf_emitter through the C ABI (src/render/tests/test_scene.py:162-201) == the oracle, entry for entry, incl. a mask"""
    from tests.test_scene_properties_cpu import _three_emitter_scene
    scene = _three_emitter_scene(mi, weights)
    osc, _ = O.scene_from_product(scene)
    pdf = np.array(weights) / np.sum(weights)
    assert np.allclose(scene.pdf_emitter([0, 1, 2]).cpu().numpy(), pdf, rtol=1e-6)
    u = ((np.arange(100001) + 0.5) / 100001).astype(np.float32)
    active = np.random.default_rng(1).random(u.size) < 0.8
    idx, w, r = scene.sample_emitter(u, active=active)
    ri, rw, rr = osc.sample_emitter(u)
    idx = idx.cpu().numpy().astype(np.uint32); w = w.cpu().numpy(); r = r.cpu().numpy()
    assert np.array_equal(idx[active], ri[active]) and np.array_equal(w[active], rw[active]) and np.array_equal(r[active], rr[active])
    assert not w[~active].any() and not r[~active].any()
    freq = np.bincount(ri, minlength=3) / u.size
    assert np.allclose(freq, pdf, atol=1e-4)


def test_weight_and_to_uv_updates_through_traverse(mi, O):
    """params['<emitter>.sampling_weight'] / params['<bsdf>.<slot>.to_uv'] + update(): the scene handle stays, the render is a freshly loaded scene's"""
    import torch
    T = mi.ScalarTransform3f
    d = two_light_scene(mi, 1.0, 1.0, res=32)
    tex = np.random.default_rng(2).uniform(0.2, 0.9, (8, 8, 3)).astype(np.float32)
    d["white"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex}}
    scene = mi.load_dict(d); mi.render(scene, spp=4, seed=0); handle = scene._h.value
    params = mi.traverse(scene)
    params["light.emitter.sampling_weight"] = torch.tensor([0.5]); params["lamp2.emitter.sampling_weight"] = torch.tensor([2.5])
    new_uv = T().rotate(30.0).scale([2.0, 3.0])
    params["white.reflectance.to_uv"] = torch.tensor(new_uv.matrix)
    params.update()
    assert scene._h.value == handle
    d2 = two_light_scene(mi, 0.5, 2.5, res=32); d2["white"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "to_uv": new_uv}}
    fresh = mi.load_dict(d2)
    a = mi.render(scene, spp=16, seed=3).cpu().numpy(); b = mi.render(fresh, spp=16, seed=3).cpu().numpy()
    assert rel_l2(a, b) < 1e-6
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=3, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(a, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices
    # back to uniform weights: the distribution goes away again
    params["light.emitter.sampling_weight"] = torch.tensor([1.0]); params["lamp2.emitter.sampling_weight"] = torch.tensor([1.0]); params.update()
    d3 = two_light_scene(mi, 1.0, 1.0, res=32); d3["white"] = d2["white"]
    c = mi.render(scene, spp=16, seed=3).cpu().numpy(); e = mi.render(mi.load_dict(d3), spp=16, seed=3).cpu().numpy()
    assert rel_l2(c, e) < 1e-6 and rel_l2(c, a) > 1e-3


def test_single_emitter_record_follows_every_kind_of_update(mi, O):
    """a scene with ONE emitter hands its record to the shading kernels as a kernel argument (DScene::emitter0): a radiance set from a device tensor, one set from the
    host, and a weight update after a device-side radiance must all reach the render -- against freshly loaded scenes"""
    import torch
    def box(radiance, weight=1.0):
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
        d["light"]["emitter"]["radiance"] = {"type": "rgb", "value": radiance}
        if weight != 1.0:
            d["light"]["emitter"]["sampling_weight"] = weight
        return d
    scene = mi.load_dict(box([18.387, 13.9873, 6.75357])); mi.render(scene, spp=4, seed=0); handle = scene._h.value
    params = mi.traverse(scene)
    key = "light.emitter.radiance.value"
    for step, (value, weight) in enumerate([(torch.tensor([4.0, 9.0, 2.0], device="cuda"), None), (torch.tensor([7.0, 1.0, 3.0]), None),
                                            (torch.tensor([2.0, 5.0, 8.0], device="cuda"), 2.0)]):
        params[key] = value
        if weight is not None:
            params["light.emitter.sampling_weight"] = torch.tensor([weight])
        params.update()
        assert scene._h.value == handle
        a = mi.render(scene, spp=16, seed=5).cpu().numpy()
        b = mi.render(mi.load_dict(box([float(x) for x in value.cpu()], weight or 1.0)), spp=16, seed=5).cpu().numpy()
        assert rel_l2(a, b) < 1e-6, step
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=5, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(a, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices


def test_bsdf_parameter_updates_keep_the_scene_handle(mi, O):
    """params['<bsdf>.alpha.value' / '.eta.value' / '.k.value' / '.specular_reflectance.value'] + update(): the record is re-lowered IN PLACE (har_scene_set_bsdf_params --
    roughplastic's transmittance table, internal reflectance and sampling weight included); the scene handle and the acceleration data survive, and the scene renders like a
    freshly loaded one with those parameters (and like the oracle).  Rounds 3-5 rebuilt the whole scene for every such update."""
    import copy
    import torch

    def rel_l2(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 48; d["sensor"]["film"]["height"] = 48
    d["metal"] = {"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}}
    d["coat"] = {"type": "roughplastic", "alpha": 0.15, "int_ior": 1.5, "diffuse_reflectance": {"type": "rgb", "value": [0.3, 0.5, 0.2]}}
    d["small-box"]["bsdf"] = {"type": "ref", "id": "metal"}; d["large-box"]["bsdf"] = {"type": "ref", "id": "coat"}
    scene = mi.load_dict(d)
    mi.render(scene, spp=4, seed=0)
    handle = scene._h.value
    params = mi.traverse(scene)
    new = {"metal.alpha.value": [0.35], "metal.eta.value": [0.4, 0.7, 1.3], "metal.k.value": [3.0, 2.0, 1.5], "coat.alpha": [0.3], "coat.specular_reflectance.value": [0.8, 0.9, 0.7]}
    assert set(new) <= set(params.keys()), sorted(params.keys())
    for k, v in new.items():
        params[k] = torch.tensor(v, device="cuda")
    params.update()
    assert scene._h is not None and scene._h.value == handle                  # nothing was rebuilt
    img = mi.render(scene, spp=32, seed=5).cpu().numpy()
    d2 = copy.deepcopy(d)
    d2["metal"].update(alpha=0.35, eta={"type": "rgb", "value": [0.4, 0.7, 1.3]}, k={"type": "rgb", "value": [3.0, 2.0, 1.5]})
    d2["coat"].update(alpha=0.3, specular_reflectance={"type": "rgb", "value": [0.8, 0.9, 0.7]})
    fresh = mi.load_dict(d2)
    ref = mi.render(fresh, spp=32, seed=5).cpu().numpy()
    assert rel_l2(img, ref) < 1e-6
    osc, sensor = O.scene_from_product(scene)
    want, ost = osc.render_path(sensor, seed=5, spp=32, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, want) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices
    # a value the plugin refuses is an error, not a silent no-op
    params["coat.alpha"] = torch.tensor([float("nan")], device="cuda")
    with pytest.raises(RuntimeError, match="not finite"):
        params.update()


def test_delta_emitter_pose_updates_keep_the_scene_handle(mi, O):
    """params['<emitter>.position'] of a point light, '.to_world' / '.cutoff_angle' / '.beam_width' of a spot light, '.to_world' of a directional light + update(): the
    records are re-lowered IN PLACE (har_scene_set_delta_emitter); the handle survives and the scene renders like a freshly loaded one and like the oracle."""
    import copy
    import torch

    def rel_l2(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
    T = mi.ScalarTransform4f
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 48; d["sensor"]["film"]["height"] = 48
    d.pop("light")
    d["bulb"] = {"type": "point", "position": [0.3, 0.2, 0.1], "intensity": {"type": "rgb", "value": [0.5, 0.4, 0.3]}}
    d["spot"] = {"type": "spot", "to_world": T().look_at(origin=[0.3, 0.9, 0.2], target=[-0.2, -1.0, 0.1], up=[0, 0, 1]), "intensity": 2.0, "cutoff_angle": 30.0, "beam_width": 20.0}
    d["sun"] = {"type": "directional", "direction": [0.2, -1.0, -0.3], "irradiance": 0.5}
    scene = mi.load_dict(d)
    mi.render(scene, spp=4, seed=0)
    handle = scene._h.value
    params = mi.traverse(scene)
    spot2 = T().look_at(origin=[-0.4, 0.8, 0.3], target=[0.3, -1.0, -0.2], up=[0, 0, 1])
    sun2 = T().look_at(origin=[0, 0, 0], target=[-0.3, -1.0, 0.2], up=[0, 0, 1])
    params["bulb.position"] = torch.tensor([-0.2, 0.4, 0.3])
    params["spot.to_world"] = torch.tensor(np.asarray(spot2.matrix, np.float32))
    params["spot.cutoff_angle"] = torch.tensor([40.0]); params["spot.beam_width"] = torch.tensor([25.0])
    params["sun.to_world"] = torch.tensor(np.asarray(sun2.matrix, np.float32))
    params.update()
    assert scene._h is not None and scene._h.value == handle
    img = mi.render(scene, spp=32, seed=5).cpu().numpy()
    d2 = copy.deepcopy(d)
    d2["bulb"]["position"] = [-0.2, 0.4, 0.3]
    d2["spot"].update(to_world=spot2, cutoff_angle=40.0, beam_width=25.0)
    d2["sun"].pop("direction"); d2["sun"]["to_world"] = sun2
    fresh = mi.load_dict(d2)
    assert rel_l2(img, mi.render(fresh, spp=32, seed=5).cpu().numpy()) < 1e-6
    osc, sensor = O.scene_from_product(scene)
    want, ost = osc.render_path(sensor, seed=5, spp=32, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, want) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices
    # an invalid cone is refused as a PAIR at the end of update(), and nothing reaches the scene
    params["spot.cutoff_angle"] = torch.tensor([10.0])
    with pytest.raises(RuntimeError, match="cutoff_angle"):
        params.update()


def test_rectangle_to_world_gradient_is_the_chain_rule_over_its_vertices(mi, O):
    """`'<rectangle>.to_world'` is Differentiable in the reference (rectangle.cpp:199).  A rectangle's four vertices ARE to_world * (+-1, +-1, 0), so  d loss / d to_world[r, c] =
    sum_v d loss / d p_v[r] * (local corner of v, 1)[c]:  held to the ORACLE's vertex-position gradients (dual numbers through prb.py:124-297) folded the same way, through
    render_backward and through mi.render + autograd; a rectangle that carries the light is refused by name; an optimiser step through params.update() moves the floor"""
    import torch
    res, spp, md = 32, 64, 4
    d, tex = textured_scene(mi, None, res=res, tex_res=8)
    d["integrator"] = {"type": "prb", "max_depth": md, "shape_gradients": ["floor.to_world"]}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(8).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    assert "floor.to_world" in grads and "floor.positions" not in grads
    mesh = scene._position_keys()["floor.positions"]
    want_v, _, _, _ = osc.render_prb_backward_shape(sensor, grad_in, [mesh], seed=3, spp=spp, max_depth=md)
    M = np.asarray(scene.meshes[mesh]["rect"]["to_world"].matrix, np.float64)
    P = np.concatenate([scene.meshes[mesh]["V"][:, :3].astype(np.float64), np.ones((4, 1))], axis=1)
    local = (np.linalg.inv(M) @ P.T).T
    assert np.allclose(np.abs(local[:, :2]), 1.0, atol=1e-5) and np.allclose(local[:, 2], 0.0, atol=1e-5)          # the corners of the unit square
    want = np.zeros((4, 4)); want[:3, :] = want_v[mesh].astype(np.float64).T @ local
    got = grads["floor.to_world"].cpu().numpy()
    scale = np.abs(want).max()
    assert scale > 0 and np.abs(got - want).max() < 2e-3 * scale and not got[3].any(), (got, want)
    # both keys at once: the positions stay in the result
    scene.integrator().shape_gradients = ["floor.to_world", "floor.positions"]
    both = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    assert np.abs(both["floor.to_world"].cpu().numpy() - got).max() < 1e-4 * scale and np.abs(both["floor.positions"].cpu().numpy() - want_v[mesh]).max() < 2e-3 * np.abs(want_v[mesh]).max()
    # the autograd route, and one descent step through params.update()
    scene.integrator().shape_gradients = False
    params = mi.traverse(scene)
    params["floor.to_world"].requires_grad_(True)
    img = mi.render(scene, params, spp=spp, seed=3)
    (img * torch.as_tensor(grad_in, device=img.device)).sum().backward()
    g = params["floor.to_world"].grad.cpu().numpy()
    assert g.shape == (4, 4) and np.isfinite(g).all() and np.abs(g).max() > 0
    before = scene.meshes[mesh]["V"][:, :3].copy()
    with torch.no_grad():
        params["floor.to_world"][1, 3] -= 0.05
    params.update()
    assert np.allclose(scene.meshes[mesh]["V"][:, 1], before[:, 1] - 0.05, atol=1e-6)
    mi.render(scene, spp=4, seed=1)
    with pytest.raises(RuntimeError, match="area light"):
        scene.integrator().shape_gradients = ["light.to_world"]
        scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=4)
    scene.integrator().shape_gradients = False
