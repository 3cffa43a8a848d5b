"""`constant` environment emitter (src/emitters/constant.cpp) next to the area light: product host pipeline vs oracle,
closed-form checks (furnace test), emitter ordering of Scene::emitters()."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def env_scene(mi, res=24, with_area=True, reflectance=0.5, radiance=(0.7, 1.1, 0.9)):
    T = mi.ScalarTransform4f
    d = mi.cornell_box() if with_area else {"type": "scene", "integrator": {"type": "path", "max_depth": 8},
                                            "sensor": mi.cornell_box()["sensor"]}
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    if with_area:
        d.pop("ceiling"); d.pop("back")                           # open the box so that rays escape
        d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": list(radiance)}}
    else:
        d["gray"] = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [reflectance] * 3}}
        d["cube"] = {"type": "cube", "to_world": T().scale([0.4, 0.4, 0.4]), "bsdf": {"type": "ref", "id": "gray"}}
        d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": list(radiance)}}
    return d


def _harness_render(mi, O, scene, sensor, mode, seed, spp, md):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    w, hgt = scene.sensors()[0].film().crop_size()
    film = np.zeros((hgt, w, 4), np.float32)
    assert L.hh_render(h, C.byref(sensor), mode, seed, spp, md, 5, 0, 0, O.fp(film)) == 0
    return film


@pytest.mark.parametrize("with_area", [True, False])
def test_constant_emitter_host_pipeline_matches_oracle(mi, O, with_area):
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    scene = mi.load_dict(env_scene(mi, 24, with_area))
    assert [e["type"] for e in scene.emitters] == ([0, 1] if with_area else [1])      # declaration order: light, then sky
    osc, sensor = oracle_scene_from(O, scene)
    for mode, md in ((0, 8), (1, 6)):
        film = _harness_render(mi, O, scene, sensor, mode, 7, 8, md)
        ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=7, spp=8, max_depth=md, raw=True, threads=2)
        assert np.isfinite(film).all() and rel_l2(O.develop(film), O.develop(ref)) < 1e-5


def test_furnace(mi, O):
    """white furnace: a closed constant environment of radiance L around a diffuse object of albedo rho: every path
    path leaves a convex object after one bounce, so the object shows rho * L and the background L (closed form: checks
    emission + emitter sampling + MIS of the environment); with rho = 1 the whole image is L."""
    from tests.test_cpu_host import oracle_scene_from
    scene = mi.load_dict(env_scene(mi, 16, with_area=False, reflectance=0.5, radiance=(1.0, 1.0, 1.0)))
    osc, sensor = oracle_scene_from(O, scene)
    img, _ = osc.render_path(sensor, seed=1, spp=256, max_depth=2)
    centre = img[6:10, 6:10].mean(axis=(0, 1)); corner = img[0, 0]
    assert np.allclose(corner, 1.0, atol=1e-5)                    # escaped camera rays see the environment
    assert np.allclose(centre, 0.5, atol=0.03)                    # one bounce on a convex object: rho * L
    # white furnace: albedo 1 -> the object is indistinguishable from the environment at any depth
    scene = mi.load_dict(env_scene(mi, 16, with_area=False, reflectance=1.0, radiance=(1.0, 1.0, 1.0)))
    osc, sensor = oracle_scene_from(O, scene)
    img, _ = osc.render_path(sensor, seed=1, spp=64, max_depth=40, rr_depth=100)
    assert np.allclose(img, 1.0, atol=0.2) and abs(float(img.mean()) - 1.0) < 0.01


def test_two_environment_emitters_are_rejected(mi):
    d = env_scene(mi, 8, True); d["sky2"] = {"type": "constant"}
    with pytest.raises(RuntimeError):
        mi.load_dict(d)
