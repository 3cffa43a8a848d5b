"""Incremental accel updates on the device (row a5 `rebuild`; src/render/scene.cpp:517-540, scene_optix.inl:351-372) through mi.traverse / params.update():
the scene handle survives, an instance edit rebuilds the instance level only, a vertex edit refits the BLAS with the kernels of har_refit.hip --
ray queries equal the brute-force kernel and a freshly loaded scene bit for bit, renders equal a fresh scene's."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def _rays(mi, n, seed=1):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.9, 0.9, (3, n)).astype(np.float32); d = rng.normal(size=(3, n)).astype(np.float32); d /= np.linalg.norm(d, axis=0)
    return mi.Ray3f(o, d.astype(np.float32))


def _pi_equal(a, b):
    import torch
    return all(torch.equal(x, y) for x, y in ((a.t, b.t), (a.prim_uv[0], b.prim_uv[0]), (a.prim_uv[1], b.prim_uv[1]), (a.prim_index, b.prim_index),
                                              (a.shape_index, b.shape_index), (a.instance, b.instance)))


def _scene_dict(mi, flatten, sky=False, res=48):
    d = mi.instanced_spheres_scene(width=res, height=res, spp=16, grid=4, n_u=40, n_v=20, flatten=flatten)
    if sky:
        d.pop("ceiling"); d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.4, 0.5, 0.6]}}
    return d


@pytest.mark.parametrize("flatten", [True, False])
def test_vertex_update_refits_in_place(mi, O, flatten):
    import torch
    d = _scene_dict(mi, flatten, sky=not flatten)
    scene = mi.load_dict(d)
    mi.render(scene, spp=4, seed=0)                                   # the handle exists
    handle = scene._h.value
    params = mi.traverse(scene)
    key = "ball005.positions" if flatten else "spheres.ball.positions"
    rays = _rays(mi, 200000)
    rng = np.random.default_rng(5)
    base = params[key].cpu().numpy().reshape(-1, 3)
    for step, amount in enumerate((0.05, 0.15, 0.3)):
        # a smooth deformation (what an optimiser does): the mesh grows and wobbles, plus a jitter well below the triangle size
        new = base * np.float32(1.0 + amount) + np.float32(0.01 * amount) * np.sin(np.float32(35.0) * base[:, ::-1]) + rng.normal(scale=2e-4, size=base.shape).astype(np.float32)
        new = np.ascontiguousarray(new, np.float32).reshape(-1)
        params[key] = torch.tensor(new, device="cuda"); params.update()
        assert scene._h is not None and scene._h.value == handle      # same scene handle: nothing was destroyed
        got = scene.ray_intersect_preliminary(rays)
        brute = scene._intersect(rays, True)
        assert int(got.is_valid().sum()) > 20000 and _pi_equal(got, brute), (step, amount)
        # a freshly loaded scene with the same geometry: same intersections, same picture, and the oracle agrees
        d2 = copy.deepcopy(d)
        m = scene._position_keys()[key]
        name = scene.meshes[m]["key"]
        assert m in scene._stale_meshes                               # a CUDA tensor: the update ran on the device, the numpy mirror was not touched
        scene.sync_host()                                             # ... until someone asks for it (positions + the normals the device regenerated)
        assert np.array_equal(scene.meshes[m]["V"][:, :3].reshape(-1), new)
        if flatten:
            d2[name]["positions"] = scene.meshes[m]["V"][:, :3].copy(); d2[name]["normals"] = scene.meshes[m]["V"][:, 3:6].copy(); d2[name].pop("to_world", None)
        else:
            d2["spheres"]["ball"]["positions"] = scene.meshes[m]["V"][:, :3].copy(); d2["spheres"]["ball"]["normals"] = scene.meshes[m]["V"][:, 3:6].copy()
        fresh = mi.load_dict(d2)
        assert _pi_equal(got, fresh.ray_intersect_preliminary(rays))
        a = mi.render(scene, spp=16, seed=3).cpu().numpy(); b = mi.render(fresh, spp=16, seed=3).cpu().numpy()
        assert rel_l2(a, b) < 1e-6
        osc, sensor = O.scene_from_product(scene)
        ref, ost = osc.render_path(sensor, seed=3, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
        assert rel_l2(a, ref) < 1e-4 and scene.integrator().stats()["vertices"] == ost.vertices
    info = scene.refit_info()
    assert info["refits"] == 3 and info["rebuilds"] == 0 and info["ratio"] >= 1.0


def test_instance_update_rebuilds_the_instance_level_only(mi, O):
    import torch
    d = _scene_dict(mi, False, sky=True)
    scene = mi.load_dict(d)
    mi.render(scene, spp=4, seed=0)
    handle = scene._h.value; nodes_before = scene.accel_info()["nodes"]
    params = mi.traverse(scene)
    T = mi.ScalarTransform4f
    rays = _rays(mi, 200000)
    moved = {"inst003.to_world": T().translate([0.3, 0.5, 0.2]).rotate([1, 0, 0], 40.0).scale(1.5),
             "inst004.to_world": T().translate([-0.3, -0.2, 0.4]).rotate([0, 1, 0], 10.0).scale(0.7),
             "inst011.to_world": T().translate([0.0, 0.1, -0.6]).scale(2.0)}
    for k, t in moved.items():
        params[k] = torch.tensor(np.asarray(t.matrix, np.float32), device="cuda")
    params.update()
    assert scene._h.value == handle
    got = scene.ray_intersect_preliminary(rays)
    assert _pi_equal(got, scene._intersect(rays, True))
    # a fresh scene with the SAME instance records: a matrix assigned through params gets a numerically inverted to_object (Instance::parameters_changed), the
    # transform chain of the dict composes analytic inverses -- both are valid, they differ in the last bit
    d2 = copy.deepcopy(d)
    scene.sync_host()                       # CUDA tensors: the transforms went to the library from the device; the Python mirror follows on request
    for k in moved:
        i = scene._instance_keys()[k]
        tw = np.asarray(scene.instances[i][1], np.float32).reshape(4, 3).T; to = np.asarray(scene.instances[i][2], np.float32).reshape(4, 3).T
        m4 = np.eye(4, dtype=np.float32); m4[:3, :] = tw; i4 = np.eye(4, dtype=np.float32); i4[:3, :] = to
        d2[k.split(".")[0]]["to_world"] = T(np.concatenate([m4.ravel(), i4.T.ravel()]))
    fresh = mi.load_dict(d2)
    for k in moved:
        i = scene._instance_keys()[k]
        assert np.array_equal(np.asarray(fresh.instances[i][1], np.float32), np.asarray(scene.instances[i][1], np.float32))
        assert np.array_equal(np.asarray(fresh.instances[i][2], np.float32), np.asarray(scene.instances[i][2], np.float32))
    assert _pi_equal(got, fresh.ray_intersect_preliminary(rays))
    a = mi.render(scene, spp=16, seed=3).cpu().numpy(); b = mi.render(fresh, spp=16, seed=3).cpu().numpy()
    assert rel_l2(a, b) < 1e-6 and abs(scene.accel_info()["nodes"] - nodes_before) <= 16
    osc, sensor = O.scene_from_product(scene)
    ref, _ = osc.render_path(sensor, seed=3, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(a, ref) < 1e-4


def test_degraded_refit_advises_a_rebuild_and_emitter_meshes_get_a_new_scene(mi):
    import torch
    d = _scene_dict(mi, True)
    scene = mi.load_dict(d); mi.render(scene, spp=4, seed=0)
    params = mi.traverse(scene)
    key = "ball002.positions"
    p = params[key].cpu().numpy().reshape(-1, 3)
    params[key] = torch.tensor(p.reshape(-1), device="cuda"); params.update()            # first refit: the baseline of the cost figure
    assert scene._h is not None and scene.refit_info()["refits"] == 1
    big = p.copy(); big[::2] += np.float32(0.6)                                        # every other vertex flies off: long thin triangles through the whole box
    params[key] = torch.tensor(big.reshape(-1), device="cuda"); params.update()
    # device-resident update: nothing waits for the refit's cost figure -- the advice arrives with the NEXT update call (include/hip_ad_rgb.h)
    assert scene._h is not None and getattr(scene, "accel_rebuilds", 0) == 0
    rays = _rays(mi, 50000)
    assert _pi_equal(scene.ray_intersect_preliminary(rays), scene._intersect(rays, True))   # valid meanwhile: a refit is exact, only slower to trace
    params[key] = torch.tensor(big.reshape(-1), device="cuda"); params.update()
    assert scene._h is None and scene.accel_rebuilds == 1                               # advised: the next render builds a fresh tree from the refreshed mirror
    assert np.array_equal(scene.meshes[scene._position_keys()[key]]["V"][:, :3], big)
    assert _pi_equal(scene.ray_intersect_preliminary(rays), scene._intersect(rays, True))
    # the same through the host path (a CPU tensor): the call waits for its own refit, the advice is immediate
    scene2 = mi.load_dict(d); mi.render(scene2, spp=4, seed=0)
    p2 = mi.traverse(scene2)
    p2[key] = torch.tensor(p.reshape(-1)); p2.update()
    assert scene2._h is not None and scene2.refit_info()["refits"] == 1
    p2[key] = torch.tensor(big.reshape(-1)); p2.update()
    assert scene2._h is None and scene2.accel_rebuilds == 1
    rays = _rays(mi, 50000)
    assert _pi_equal(scene.ray_intersect_preliminary(rays), scene._intersect(rays, True))
    # a mesh with an area emitter: its sampling records are lowered from the positions -> new scene, not a refit
    cb = mi.load_dict(mi.cornell_box()); mi.render(cb, spp=4, seed=0)
    pc = mi.traverse(cb)
    pc["light.positions"] = pc["light.positions"] * 1.0; pc.update()
    assert cb._h is None


def test_shape_optimisation_steps_keep_the_handle(mi):
    """vertex positions through mi.render + autograd + an optimiser step + params.update(): the loop of examples/optimize_vertices.py, three steps"""
    import torch
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["integrator"] = {"type": "prb", "max_depth": 4}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = "small-box.positions"
    params[key] = params[key].clone().requires_grad_(True); params.update()
    opt = torch.optim.SGD([params[key]], lr=1e-3)
    mi.render(scene, spp=4, seed=0); handle = scene._h.value
    for it in range(3):
        opt.zero_grad()
        img = mi.render(scene, params, spp=16, seed=it)
        (img ** 2).mean().backward()
        assert torch.isfinite(params[key].grad).all() and float(params[key].grad.abs().max()) > 0
        opt.step(); params.update()
        assert scene._h is not None and scene._h.value == handle
    assert scene.refit_info()["refits"] == 3



@pytest.mark.parametrize("flatten,sky", [(True, False), (False, True), (False, False)], ids=["top-level", "instanced+sky", "instanced"])
def test_device_resident_vertex_update_equals_the_host_update(mi, O, flatten, sky):
    """params[key] as a CUDA tensor (har_scene_update_vertices_device: positions, regenerated normals, shading triangles, refit -- all kernels) against the same
    values as a CPU tensor (har_mesh_compute_normals on the host + har_scene_update_vertices): same intersections bit for bit, same picture, same vertex records;
    "instanced+sky" carries a `constant` emitter, so the instance level and the scene's bounding sphere follow on the host (the call's read-back path); "instanced" has the
    instance level REFITTED on the device (k_instance_boxes + the TLAS levels: no read-back, no wait) where the host path rebuilds it -- same answers, boxes only prune."""
    import torch
    d = _scene_dict(mi, flatten, sky=sky)
    key = "ball005.positions" if flatten else "spheres.ball.positions"
    a = mi.load_dict(d); b = mi.load_dict(copy.deepcopy(d))
    for sc in (a, b):
        mi.render(sc, spp=4, seed=0)
    pa = mi.traverse(a); pb = mi.traverse(b)
    base = pa[key].cpu().numpy().reshape(-1, 3)
    rays = _rays(mi, 100000, seed=2)
    m = a._position_keys()[key]
    mirror_before = a.meshes[m]["V"].copy()
    for amount in (0.08, 0.2):
        new = np.ascontiguousarray(base * np.float32(1.0 + amount) + np.float32(0.01) * np.sin(np.float32(30.0) * base[:, ::-1]), np.float32).reshape(-1)
        pa[key] = torch.tensor(new, device="cuda"); pa.update()
        pb[key] = torch.tensor(new); pb.update()                      # CPU tensor -> host path
        assert np.array_equal(a.meshes[m]["V"], mirror_before)        # the device path never wrote (or read) the host mirror
        ga = a.ray_intersect_preliminary(rays); gb = b.ray_intersect_preliminary(rays)
        assert int(ga.is_valid().sum()) > 10000 and _pi_equal(ga, gb) and _pi_equal(ga, a._intersect(rays, True))
        ia = mi.render(a, spp=16, seed=5).cpu().numpy(); ib = mi.render(b, spp=16, seed=5).cpu().numpy()
        assert rel_l2(ia, ib) < 1e-6
    a.sync_host()
    Va, Vb = a.meshes[m]["V"], b.meshes[m]["V"]
    assert np.array_equal(Va[:, :3], Vb[:, :3]) and np.array_equal(Va[:, 6:], Vb[:, 6:])
    # regenerated normals: the device gathers a vertex's corner terms in the serial loop's order, asinf / sqrt are the device's -> equal to a few ulps
    assert np.abs(Va[:, 3:6] - Vb[:, 3:6]).max() < 2e-6
    assert a.refit_info()["refits"] == 2 and a.refit_info()["rebuilds"] == 0


def test_device_resident_update_reports_a_non_finite_position_one_call_late(mi):
    import torch
    d = _scene_dict(mi, True)
    scene = mi.load_dict(d); mi.render(scene, spp=4, seed=0)
    params = mi.traverse(scene)
    key = "ball002.positions"
    p = params[key].clone()
    bad = p.clone(); bad[7] = float("nan")
    params[key] = bad; params.update()                                # enqueued; nothing has looked at the values yet
    assert scene._h is not None
    params[key] = p.clone()
    with pytest.raises(RuntimeError, match="not finite"):
        params.update()
    assert scene._h is None                                           # the handle that held the bad geometry is gone; the next render builds from the mirror
    params.update()                                                   # the rejected key is looked at again (no scene handle: the host path sets the good positions)
    rays = _rays(mi, 20000)
    assert _pi_equal(scene.ray_intersect_preliminary(rays), scene._intersect(rays, True))


def test_vertex_loop_with_device_updates_matches_host_updates(mi):
    """three optimiser steps over vertex positions on the GPU: mi.render + backward + SGD + params.update() with device-resident updates == the same loop forced
    through the host path (HAR_HOST_VERTEX_UPDATE=1), gradients and positions"""
    import os
    import torch

    def loop(host):
        if host:
            os.environ["HAR_HOST_VERTEX_UPDATE"] = "1"
        try:
            from mitsuba3_amd.scenes import bumpy_sphere
            d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
            d["integrator"] = {"type": "prb", "max_depth": 4}
            P, N, UV, F = bumpy_sphere(n_u=32, n_v=16, radius=0.35)
            d.pop("small-box"); d.pop("large-box")
            d["blob"] = {"type": "mesh", "positions": P + np.array([0.0, -0.45, 0.0], np.float32), "normals": N, "faces": F, "bsdf": {"type": "ref", "id": "white"}}
            scene = mi.load_dict(d)
            params = mi.traverse(scene)
            key = "blob.positions"
            params[key] = params[key].clone().requires_grad_(True); params.update()
            opt = torch.optim.SGD([params[key]], lr=2e-7)               # gradients of mean(img^2) w.r.t. a vertex reach ~1e3: steps of ~1e-4 scene units
            grads = []
            for it in range(3):
                opt.zero_grad()
                img = mi.render(scene, params, spp=16, seed=it)
                (img ** 2).mean().backward()
                grads.append(params[key].grad.detach().cpu().numpy().copy())
                opt.step(); params.update()
            return grads, params[key].detach().cpu().numpy().copy(), scene
        finally:
            os.environ.pop("HAR_HOST_VERTEX_UPDATE", None)

    gd, pd, sd = loop(False); gh, ph, sh = loop(True)
    assert sd.device_vertex_updates == 3 and getattr(sh, "device_vertex_updates", 0) == 0
    for a, b in zip(gd, gh):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert rel_l2(a, b) < 1e-3                                    # atomics order only (first step: identical geometry)
    assert np.abs(pd - ph).max() < 1e-6


@pytest.mark.parametrize("sky", [False, True], ids=["device", "sky: host fallback"])
def test_device_resident_instance_update(mi, O, sky):
    """params['<instance>.to_world'] as CUDA tensors (har_scene_update_instances_device: inverses, shading / TLAS leaf records, instance bounds and the REFIT of the instance level
    are kernels; nothing is copied or waited for) against the same matrices through the host path (np.linalg.inv + har_scene_update_instances: the instance level is rebuilt):
    the scene answers like the brute-force kernel bit for bit, and like the host-path scene up to the last bit of the two inverses (picture 1e-5).  With a `constant` emitter the
    scene's bounding sphere follows the instances: the call reads the matrices back and takes the host path."""
    import os
    import torch
    d = _scene_dict(mi, False, sky=sky)
    T = mi.ScalarTransform4f
    moved = {"inst003.to_world": T().translate([0.3, 0.5, 0.2]).rotate([1, 0, 0], 40.0).scale(1.5),
             "inst004.to_world": T().translate([-0.3, -0.2, 0.4]).rotate([0, 1, 0], 10.0).scale(0.7),
             "inst011.to_world": T().translate([0.0, 0.1, -0.6]).scale(2.0)}
    a = mi.load_dict(d); b = mi.load_dict(copy.deepcopy(d))
    for sc in (a, b):
        mi.render(sc, spp=4, seed=0)
    handle = a._h.value
    pa = mi.traverse(a); pb = mi.traverse(b)
    mirror_before = [list(x[1]) for x in a.instances]
    for k, t in moved.items():
        pa[k] = torch.tensor(np.asarray(t.matrix, np.float32), device="cuda")
        pb[k] = torch.tensor(np.asarray(t.matrix, np.float32))                       # CPU tensor: host path
    pa.update(); pb.update()
    assert a._h.value == handle
    if not sky:
        assert a.device_instance_updates == 2 and a._stale_instances                 # two runs of consecutive instances (3-4, 11), mirror untouched
        assert [list(x[1]) for x in a.instances] == mirror_before
    rays = _rays(mi, 200000, seed=4)
    ga = a.ray_intersect_preliminary(rays)
    assert int(ga.is_valid().sum()) > 20000 and _pi_equal(ga, a._intersect(rays, True))
    gb = b.ray_intersect_preliminary(rays)
    same = (ga.is_valid() == gb.is_valid()).float().mean().item()
    assert same > 0.9999                                                             # the two inverses differ in the last bit at most: grazing rays may flip
    ia = mi.render(a, spp=16, seed=3).cpu().numpy(); ib = mi.render(b, spp=16, seed=3).cpu().numpy()
    assert rel_l2(ia, ib) < 1e-5
    a.sync_host()
    for k in moved:
        i = a._instance_keys()[k]
        assert np.allclose(np.asarray(a.instances[i][1], np.float32), np.asarray(b.instances[i][1], np.float32), rtol=0, atol=0)
        assert np.allclose(np.asarray(a.instances[i][2], np.float32), np.asarray(b.instances[i][2], np.float32), rtol=1e-6, atol=1e-7)
    if not sky:
        # a singular matrix keeps the instance where it was and is reported by the NEXT update
        pa["inst003.to_world"] = torch.zeros((4, 4), device="cuda"); pa.update()
        assert a._h is not None
        pa["inst004.to_world"] = torch.tensor(np.asarray(moved["inst004.to_world"].matrix, np.float32), device="cuda")
        with pytest.raises(RuntimeError, match="singular or not finite"):
            pa.update()
