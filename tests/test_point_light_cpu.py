"""`point` emitter (src/emitters/point.cpp; HarEmitter type 4): the oracle and the product's host-compiled shading code against
  * the closed form the reference's own test checks (src/emitters/tests/test_point.py:62-121: pdf 1, delta, d = normalised offset, value = intensity / dist^2)
    through the one observable a render offers: radiance of a diffuse plane = albedo / pi * cos(theta) * I / dist^2 per camera sample,
  * src/render/tests/test_ad.py:6-95 LITERALLY (its scene: a rectangle of albedo 0.6 under a point light at (0, 0, 5), 1 x 1 pixel, box filter, the default
    integrator depth; spp 1 / 4 / 44): one gradient-descent step on a linear function, loss(rho + lr) == loss(rho) + lr * d loss / d rho,
  * each other (oracle == product host code, forward and prb)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


@pytest.fixture(scope="module")
def H(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p
    L.hh_scene_destroy.argtypes = [C.c_void_p]
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p]
    return L


def _harness_scene(H, scene):
    d = scene.desc(); err = C.create_string_buffer(256)
    h = C.c_void_p(H.hh_scene_create(C.byref(d), err, 256))
    assert h, err.value
    return h


def simple_scene(mi, res=1, integrator="path"):
    """make_simple_scene of src/render/tests/test_ad.py:6-43, key for key"""
    return {
        'type': 'scene',
        "integrator": {"type": integrator},
        "mysensor": {
            "type": "perspective", "near_clip": 0.1, "far_clip": 1000.0,
            "to_world": mi.ScalarTransform4f().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
            "myfilm": {"type": "hdrfilm", "rfilter": {"type": "box"}, "width": res, "height": res},
            "mysampler": {"type": "independent", "sample_count": 1},
        },
        'rect': {'type': 'rectangle', "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.6, 0.6, 0.6]}}},
        "emitter": {"type": "point", "position": [0, 0, 5]},
    }


def test_point_light_plugin_properties(mi):
    """point.cpp:64-77: `position` or the translation of `to_world`, not both; default intensity 1; traverse exposes `intensity` (:84-88)"""
    s = mi.load_dict(simple_scene(mi))
    e = s.emitters[0]
    assert e["type"] == 4 and list(e["to_world"][9:12]) == [0.0, 0.0, 5.0] and np.allclose(e["radiance"], 1.0)
    d = simple_scene(mi); d["emitter"] = {"type": "point", "to_world": mi.ScalarTransform4f().translate([1.0, -2.0, 3.0]), "intensity": {"type": "rgb", "value": [1.0, 2.0, 3.0]}}
    e = mi.load_dict(d).emitters[0]
    assert np.allclose(e["to_world"][9:12], [1.0, -2.0, 3.0]) and np.allclose(e["radiance"], [1.0, 2.0, 3.0])
    d["emitter"]["position"] = [0, 0, 1]
    with pytest.raises(RuntimeError, match="Only one of the parameters 'position' and 'to_world'"):
        mi.load_dict(d)
    d = simple_scene(mi); d["emitter"]["radiance"] = 1.0
    with pytest.raises(RuntimeError, match="Unreferenced property"):
        mi.load_dict(d)
    assert "emitter.intensity.value" in mi.load_dict(simple_scene(mi))._param_keys()


@pytest.mark.parametrize("spp", [1, 4, 64])
def test_closed_form_radiance_under_a_point_light(mi, O, spp):
    """every camera sample that meets the rectangle returns albedo / pi * cos(theta) / dist^2 (unit intensity, nothing occludes, the BSDF-sampled ray escapes into
    darkness); the others return 0: the pixel is the mean over the oracle's own camera rays"""
    scene = mi.load_dict(simple_scene(mi))
    osc, sensor = O.scene_from_product(scene)
    img, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=-1)
    # the same camera rays, from the oracle's sampler stream: pixel jitter = the first two floats of lane i (sampler.cpp:129-148, integrator.cpp:322-345)
    px = np.zeros(spp, np.float32); py = np.zeros(spp, np.float32)
    for i in range(spp):
        s2 = np.zeros(2, np.float32); O.lib().orc_sampler_stream(0, i, 2, O.fp(s2)); px[i], py[i] = s2
    o = np.zeros((3, spp), np.float32); d = np.zeros((3, spp), np.float32); mt = np.zeros(spp, np.float32)
    O.lib().orc_sensor_sample_ray(C.byref(sensor), spp, O.fp(px), O.fp(py), O.fp(o), O.fp(d), O.fp(mt))
    t = -o[2] / d[2]; p = o + t * d
    hit = (np.abs(p[0]) <= 1) & (np.abs(p[1]) <= 1)
    L = np.array([0.0, 0.0, 5.0])[:, None] - p
    dist2 = (L ** 2).sum(0); cos = L[2] / np.sqrt(dist2)
    ref = np.where(hit, 0.6 / np.pi * cos / dist2, 0.0).mean()
    assert hit.any()
    assert np.allclose(img[0, 0], ref, rtol=2e-5), (img[0, 0], ref)
    assert st.shadow_rays == int(hit.sum())          # one emitter sample per vertex, none for the samples that miss


@pytest.mark.parametrize("spp", [1, 4, 44])
def test01_bsdf_reflectance_backward_literal(mi, O, spp):
    """src/render/tests/test_ad.py:55-95 (its scene, its seed, its spp values, its identity; the adjoint comes from the oracle's prb)"""
    scene = mi.load_dict(simple_scene(mi))
    osc, sensor = O.scene_from_product(scene)
    img1, _ = osc.render_prb(sensor, seed=0, spp=spp, max_depth=-1)
    grad_in = np.ones((1, 1, 3), np.float32)                                  # loss = sum(img)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=0, spp=spp, max_depth=-1)
    lr = 0.01
    osc.set_reflectance(0, np.array([0.6 + lr, 0.6, 0.6], np.float32))
    img2, _ = osc.render_prb(sensor, seed=0, spp=spp, max_depth=-1)
    assert img1.sum() > 0 or spp == 1
    assert np.isclose(img1.sum(), img2.sum() - lr * g_refl[0, 0], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_product_host_shading_matches_oracle_with_point_lights(mi, O, H, mode, md):
    """the Cornell box lit by its area light AND two point lights (three emitters: the uniform emitter choice and sample re-use of scene.cpp:248-271), through
    the product's shade_lane on the host against the oracle, path and prb"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["bulb"] = {"type": "point", "position": [0.3, 0.2, 0.1], "intensity": {"type": "rgb", "value": [0.5, 0.4, 0.3]}}
    d["bulb2"] = {"type": "point", "to_world": mi.ScalarTransform4f().translate([-0.5, -0.4, 0.5]), "intensity": 0.2}
    scene = mi.load_dict(d)
    assert [e.get("type", 0) for e in scene.emitters].count(4) == 2
    osc, sensor = O.scene_from_product(scene)
    h = _harness_scene(H, scene)
    film = np.zeros((32, 32, 4), np.float32)
    assert H.hh_render(h, C.byref(sensor), mode, 4, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)
    # the point lights matter: without them the picture is a different one
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = 32; d2["sensor"]["film"]["height"] = 32
    osc2, s2 = O.scene_from_product(mi.load_dict(d2))
    ref2, _ = (osc2.render_path if mode == 0 else osc2.render_prb)(s2, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(ref2), O.develop(ref)) > 0.05


def test_oracle_intensity_gradient_is_the_finite_difference(mi, O):
    """PointLightIntensityConfig's parameter (test_ad_integrators.py:348-367): the image is linear in the intensity, so the prb gradient equals a finite difference of the
    primal render to rounding"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 16; d["sensor"]["film"]["height"] = 16
    d["bulb"] = {"type": "point", "position": [0.3, 0.2, 0.1], "intensity": {"type": "rgb", "value": [0.5, 0.4, 0.3]}}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    k = [e.get("type", 0) for e in scene.emitters].index(4)
    grad_in = np.ones((16, 16, 3), np.float32)
    _, _, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=2, spp=16, max_depth=4)
    a, _ = osc.render_prb(sensor, seed=2, spp=16, max_depth=4)
    osc.set_emitter_radiance(k, np.array([1.5, 0.4, 0.3], np.float32))
    b, _ = osc.render_prb(sensor, seed=2, spp=16, max_depth=4)
    fd = (b.astype(np.float64).sum() - a.astype(np.float64).sum()) / 1.0
    assert g_emit[k, 0] > 0 and np.isclose(g_emit[k, 0], fd, rtol=2e-4)


def test_point_light_from_xml(mi):
    """the XML form of the plugin's documentation (point.cpp:40-47): <point name="position"/> + <rgb name="intensity"/>"""
    s = mi.load_string('<scene version="3.0.0"><shape type="rectangle" id="rect"><bsdf type="diffuse"/></shape>'
                       '<emitter type="point" id="bulb"><point name="position" value="0.0, 5.0, 0.0"/><rgb name="intensity" value="1.0, 2.0, 3.0"/></emitter></scene>')
    e = s.emitters[0]
    assert e["type"] == 4 and list(e["to_world"][9:12]) == [0.0, 5.0, 0.0] and np.allclose(e["radiance"], [1.0, 2.0, 3.0])
    assert "bulb.intensity.value" in s._param_keys()
