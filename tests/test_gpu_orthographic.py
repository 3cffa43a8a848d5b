"""`orthographic` sensor on the device (k_raygen / har_sensor_sample_ray with HarSensor::projection = 1) against the oracle, and the perspective camera once more
behind the same kernels (the ray generation gained a branch)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def ortho_box(mi, res):
    d = mi.cornell_box()
    film = dict(d["sensor"]["film"]); film["width"] = res; film["height"] = res
    d["sensor"] = {"type": "orthographic", "near_clip": 0.01, "far_clip": 100.0, "film": film,
                   "to_world": mi.ScalarTransform4f().look_at(origin=[0.1, 0.05, 3.9], target=[0, 0, 0], up=[0, 1, 0]) @ mi.ScalarTransform4f().scale([0.9, 0.9, 1.0])}
    return d


def test_sample_ray_matches_the_oracle(mi, O):
    """OrthographicCamera::sample_ray (orthographic.cpp:131-157) through har_sensor_sample_ray: origins, directions and maxt are the oracle's bits"""
    scene = mi.load_dict(ortho_box(mi, 64))
    cam = scene.sensors()[0]
    rng = np.random.default_rng(2); n = 10000
    pos = rng.uniform(0, 1, (2, n)).astype(np.float32)
    ray, w = cam.sample_ray(0.0, 0.0, pos)
    s = O.Sensor(); C.memmove(C.byref(s), C.byref(cam.har), C.sizeof(s))
    o = np.zeros((3, n), np.float32); d = np.zeros((3, n), np.float32); mt = np.zeros(n, np.float32)
    O.lib().orc_sensor_sample_ray(C.byref(s), n, O.fp(np.ascontiguousarray(pos[0])), O.fp(np.ascontiguousarray(pos[1])), O.fp(o), O.fp(d), O.fp(mt))
    assert np.array_equal(ray.o.cpu().numpy(), o) and np.array_equal(ray.d.cpu().numpy(), d) and np.array_equal(ray.maxt.cpu().numpy(), mt)
    assert np.allclose(d, d[:, :1])                                   # parallel rays


@pytest.mark.parametrize("spp", [16, 64])
def test_forward_parity_through_an_orthographic_camera(mi, O, spp):
    """spp = 64: the camera rays go through the wave-shared descent (one origin interval per packet instead of one origin)"""
    res = 64
    scene = mi.load_dict(ortho_box(mi, res))
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=5).cpu().numpy()
    st = scene.integrator().stats()
    ref, ost = osc.render_path(sensor, seed=5, spp=spp, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert np.abs(ref).max() > 0 and rel_l2(img, ref) < 1e-4, rel_l2(img, ref)
    assert st["paths"] == res * res * spp and st["vertices"] == ost.vertices


def test_prb_gradients_through_an_orthographic_camera(mi, O):
    res, spp, md = 48, 16, 6
    d = mi.textured_cornell_box(res=res, tex_res=16, spp=4)
    film = dict(d["sensor"]["film"])
    d["sensor"] = {"type": "orthographic", "near_clip": 0.01, "far_clip": 100.0, "film": film,
                   "to_world": mi.ScalarTransform4f().look_at(origin=[0.1, 0.05, 3.9], target=[0, 0, 0], up=[0, 1, 0]) @ mi.ScalarTransform4f().scale([0.9, 0.9, 1.0])}
    d["integrator"] = {"type": "prb", "max_depth": md}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(7).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=md)
    for k, (kind, b) in scene._param_keys().items():
        ref = g_emit[b] if kind == "emit" else (g_tex[b.tex_index] if kind == "tex" else g_refl[b.index])
        assert rel_l2(grads[k].cpu().numpy().reshape(-1), np.asarray(ref).reshape(-1)) < 1e-3, k


def test_perspective_camera_unchanged(mi, O):
    """the same kernels, projection = 0: the Cornell box as every other test renders it"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 64; d["sensor"]["film"]["height"] = 64
    scene = mi.load_dict(d)
    assert scene.sensors()[0].har.projection == 0
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=64, seed=2).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=2, spp=64, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
