"""Emitter::sample_direction of the delta emitters (`point`, `spot`, `directional`) taken ALONE -- orc_emitter_sample_direction (oracle) and hh_emitter_sample_direction (the
product's shading headers compiled for the host) -- against the expectations of the reference's own emitter tests, computed as those tests compute them:
  * src/emitters/tests/test_spot.py:38-89   test_sample_direction: both look-at transforms, both cut-off angles (20 / 80 degrees), both positions -- pdf 1, delta,
    d = the normalised offset, value = intensity * falloff(angle) / dist^2 with the linear transition between beam width (3/4 of the cut-off) and cut-off;
  * src/emitters/tests/test_point.py:62-121  test02 / test03: pdf 1, delta, d, intensity / dist^2 (scalar and three positions at once);
  * src/emitters/tests/test_directional.py:40-62,88-118  test_construct (identity by default; `direction` (0, 0, -1) gives the matrix written there) and
    test_sample_direction (d = -direction / |direction| for three directions, pdf 1, the irradiance without attenuation);
and against each other (oracle == product host code, bit for bit on random positions), plus renders of a scene lit by a spot light (product host code == oracle)."""
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


def sample_direction(O, H, mi, scene_dict, index, p, s=None):
    """-> {"oracle": (d, dist, pdf, delta, weight), "product": (...)} for the emitter `index` of the scene"""
    scene = mi.load_dict(scene_dict)
    p = np.ascontiguousarray(np.asarray(p, np.float32).reshape(-1, 3)); n = p.shape[0]
    s = np.ascontiguousarray(np.zeros((n, 2), np.float32) if s is None else np.asarray(s, np.float32).reshape(n, 2))
    out = {}
    osc, _ = O.scene_from_product(scene)
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    for name, fn, handle in (("oracle", O.lib().orc_emitter_sample_direction, osc.handle), ("product", H.hh_emitter_sample_direction, h)):
        d = np.zeros((n, 3), np.float32); dist = np.zeros(n, np.float32); pdf = np.zeros(n, np.float32); delta = np.zeros(n, np.uint8); w = np.zeros((n, 3), np.float32)
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p, O.c_f32p, O.c_f32p, C.POINTER(C.c_uint8), O.c_f32p]
        fn(handle, index, n, O.fp(p), O.fp(s), O.fp(d), O.fp(dist), O.fp(pdf), delta.ctypes.data_as(C.POINTER(C.c_uint8)), O.fp(w))
        out[name] = (d, dist, pdf, delta, w)
    H.hh_scene_destroy(h)
    return out


def base_scene(mi):
    return {"type": "scene", "rect": {"type": "rectangle", "bsdf": {"type": "diffuse"}},
            "sensor": {"type": "perspective", "to_world": mi.ScalarTransform4f().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                       "film": {"type": "hdrfilm", "width": 8, "height": 8}}}


LOOKATS = [([0, 1, 0], [0, 0, 0], [1, 0, 0]), ([0, 0, 1], [0, 0, 0], [0, -1, 0])]      # the two matrices of test_spot.py:16-19, by their own comments


@pytest.mark.parametrize("it_pos", [[2.0, 0.5, 0.0], [1.0, 0.5, -5.0]])
@pytest.mark.parametrize("cutoff_angle", [20, 80])
@pytest.mark.parametrize("lookat", [0, 1])
def test_spot_sample_direction(mi, O, H, it_pos, cutoff_angle, lookat):
    """src/emitters/tests/test_spot.py:38-89"""
    origin, target, up = LOOKATS[lookat]
    T = mi.ScalarTransform4f().look_at(origin=origin, target=target, up=up)
    M = np.eye(4); M[:3, :] = np.asarray(T.col_major_3x4(), np.float64).reshape(4, 3).T
    ref_m = [np.array([[0, 1, 0, 0], [0, 0, -1, 1], [-1, 0, 0, 0], [0, 0, 0, 1]], float), np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 1], [0, 0, 0, 1]], float)][lookat]
    assert np.allclose(M, ref_m, atol=1e-6)                       # look_at gives the matrix the reference's test writes down
    cutoff_rad = cutoff_angle / 180 * np.pi; beam_rad = cutoff_rad * 0.75; inv_tw = 1 / (cutoff_rad - beam_rad)
    d = base_scene(mi); d["spot"] = {"type": "spot", "cutoff_angle": cutoff_angle, "to_world": T, "intensity": {"type": "rgb", "value": [1.0, 2.0, 0.5]}}
    res = sample_direction(O, H, mi, d, 0, it_pos)
    p = np.asarray(it_pos, np.float64)
    dd = -p + M[:3, 3]; dist = np.linalg.norm(dd); dd /= dist
    angle = np.arccos((np.linalg.inv(M)[:3, :3] @ (-dd))[2])
    if abs(angle - beam_rad) < 1e-3: angle = beam_rad
    if abs(angle - cutoff_rad) < 1e-3: angle = cutoff_rad
    spec = np.array([1.0, 2.0, 0.5])
    spec = spec if angle <= beam_rad else spec * ((cutoff_rad - angle) * inv_tw)
    spec = spec if angle <= cutoff_rad else spec * 0
    for name, (gd, gdist, gpdf, gdelta, gw) in res.items():
        assert gpdf[0] == 1.0 and gdelta[0] == 1, name
        assert np.allclose(gd[0], dd, rtol=1e-5, atol=1e-7), name
        assert np.allclose(gw[0], spec / dist ** 2, rtol=1e-5, atol=1e-8), (name, gw[0], spec / dist ** 2)


def test_point_sample_direction(mi, O, H):
    """src/emitters/tests/test_point.py:62-121 (test02: one position, emitter at (10, -1, 2); test03: three positions, emitter at (50, -1, 2))"""
    for pos, pts in (([10, -1, 2], [[0.0, -2.0, 4.5]]), ([50, -1, 2], [[0.0, 0.0, 0.0], [-2.0, 0.0, -2.0], [4.5, 4.5, 0.0]])):
        d = base_scene(mi); d["bulb"] = {"type": "point", "position": pos, "intensity": {"type": "rgb", "value": [0.7, 1.0, 1.3]}}
        res = sample_direction(O, H, mi, d, 0, pts, s=[[0.1, 0.5]] * len(pts))
        P = np.asarray(pts, np.float64); dd = -P + np.asarray(pos, np.float64); dist = np.linalg.norm(dd, axis=1); dd /= dist[:, None]
        for name, (gd, gdist, gpdf, gdelta, gw) in res.items():
            assert np.all(gpdf == 1.0) and np.all(gdelta == 1), name
            assert np.allclose(gd, dd, atol=1e-3 if len(pts) == 3 else 1e-6), name                      # the reference's own tolerances
            assert np.allclose(gw, np.array([0.7, 1.0, 1.3])[None, :] / dist[:, None] ** 2, rtol=1e-5), name
            assert np.allclose(gdist, dist, rtol=1e-6), name


def test_oracle_and_product_agree_bit_for_bit(mi, O, H):
    """random reference points around a spot light, a point light, the Cornell box's area light and a constant environment: the two implementations of
    Emitter::sample_direction return the same bits"""
    rng = np.random.default_rng(3)
    P = rng.uniform(-3, 3, (4000, 3)).astype(np.float32); S = rng.uniform(0, 1, (4000, 2)).astype(np.float32)
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 8; d["sensor"]["film"]["height"] = 8
    d["spot"] = {"type": "spot", "cutoff_angle": 35.0, "beam_width": 10.0, "intensity": {"type": "rgb", "value": [3.0, 2.0, 1.0]},
                 "to_world": mi.ScalarTransform4f().look_at(origin=[0.3, 0.9, 0.2], target=[-0.2, -1.0, 0.1], up=[0, 0, 1])}
    d["bulb"] = {"type": "point", "position": [0.3, 0.2, 0.1], "intensity": 0.5}
    d["sky"] = {"type": "constant", "radiance": 0.1}
    d["sun"] = {"type": "directional", "direction": [0.3, -1.0, -0.2], "irradiance": {"type": "rgb", "value": [2.0, 1.5, 1.0]}}
    scene = mi.load_dict(d)
    kinds = [e.get("type", 0) for e in scene.emitters]
    assert sorted(kinds) == [0, 1, 4, 5, 6]
    lit = 0
    for k in range(len(kinds)):
        res = sample_direction(O, H, mi, d, k, P, S)
        for a, b in zip(res["oracle"], res["product"]):
            assert np.array_equal(a, b), (kinds[k],)
        lit += int((res["oracle"][4].sum(1) > 0).sum())
        if kinds[k] == 5:
            w = res["oracle"][4].sum(1)
            assert 0 < (w > 0).sum() < len(w)                 # some points inside the cone, some outside
    assert lit > 0


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_product_host_shading_matches_oracle_with_a_spot_light(mi, O, H, mode, md):
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["spot"] = {"type": "spot", "cutoff_angle": 40.0, "intensity": {"type": "rgb", "value": [3.0, 2.0, 1.0]},
                 "to_world": mi.ScalarTransform4f().look_at(origin=[0.3, 0.9, 0.2], target=[-0.2, -1.0, 0.1], up=[0, 0, 1])}
    scene = mi.load_dict(d)
    assert "spot.intensity.value" in scene._param_keys()
    osc, sensor = O.scene_from_product(scene)
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((32, 32, 4), np.float32)
    assert H.hh_render(h, C.byref(sensor), mode, 4, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = 32; d2["sensor"]["film"]["height"] = 32
    osc2, s2 = O.scene_from_product(mi.load_dict(d2))
    ref2, _ = (osc2.render_path if mode == 0 else osc2.render_prb)(s2, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(ref2), O.develop(ref)) > 0.02


def test_spot_plugin_properties(mi):
    """spot.cpp:90-108: defaults (cutoff 20, beam width 3/4 of it, intensity 1), `texture` refused, unknown properties refused"""
    d = base_scene(mi); d["spot"] = {"type": "spot"}
    e = mi.load_dict(d).emitters[0]
    assert e["type"] == 5 and e["normal"][0] == 20.0 and e["normal"][1] == 15.0 and np.allclose(e["radiance"], 1.0)
    d["spot"] = {"type": "spot", "texture": {"type": "bitmap"}}
    with pytest.raises(RuntimeError, match="texture"):
        mi.load_dict(d)
    d["spot"] = {"type": "spot", "radius": 1.0}
    with pytest.raises(RuntimeError, match="Unreferenced property"):
        mi.load_dict(d)
    d["spot"] = {"type": "spot", "cutoff_angle": 10.0, "beam_width": 20.0}
    with pytest.raises(RuntimeError, match="cutoff_angle"):
        mi.load_dict(d)


def test_directional_construct(mi):
    """src/emitters/tests/test_directional.py:40-62"""
    d = base_scene(mi); d["sun"] = {"type": "directional"}
    e = mi.load_dict(d).emitters[0]
    assert e["type"] == 6 and np.allclose(np.asarray(e["to_world"]).reshape(4, 3).T, np.eye(4)[:3])
    d["sun"] = {"type": "directional", "direction": [0, 0, -1]}
    M = np.asarray(mi.load_dict(d).emitters[0]["to_world"]).reshape(4, 3).T
    assert np.allclose(M, [[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, -1, 0]], atol=1e-7)
    d["sun"] = {"type": "directional", "direction": [0, 0, -1], "to_world": mi.ScalarTransform4f()}
    with pytest.raises(RuntimeError, match="Only one of the parameters 'direction' and 'to_world'"):
        mi.load_dict(d)
    assert "sun.irradiance.value" in mi.load_dict({**base_scene(mi), "sun": {"type": "directional"}})._param_keys()


@pytest.mark.parametrize("direction", [[0, 0, -1], [1, 1, 1], [0, 0, 1]])
def test_directional_sample_direction(mi, O, H, direction):
    """src/emitters/tests/test_directional.py:88-118"""
    d = base_scene(mi); d["sun"] = {"type": "directional", "direction": direction, "irradiance": {"type": "rgb", "value": [0.7, 1.0, 1.3]}}
    res = sample_direction(O, H, mi, d, 0, [[-0.5, 0.3, -0.1]], s=[[0.85, 0.13]])
    want = -np.asarray(direction, np.float64) / np.linalg.norm(direction)
    for name, (gd, gdist, gpdf, gdelta, gw) in res.items():
        assert np.allclose(gd[0], want, atol=1e-6), name
        assert gpdf[0] == 1.0 and gdelta[0] == 1, name
        assert np.allclose(gw[0], [0.7, 1.0, 1.3]), name             # no attenuation with distance
        assert gdist[0] > 0


def test_closed_form_radiance_under_a_directional_light(mi, O):
    """a diffuse rectangle (albedo 0.5) under irradiance E arriving at 60 degrees from its normal: every camera sample on it returns 0.5 / pi * E * cos(60 deg)"""
    dirn = np.array([np.sin(np.pi / 3), 0.0, -np.cos(np.pi / 3)])
    d = {"type": "scene", "rect": {"type": "rectangle", "to_world": mi.ScalarTransform4f().scale(10.0), "bsdf": {"type": "diffuse"}},
         "sun": {"type": "directional", "direction": [float(x) for x in dirn], "irradiance": {"type": "rgb", "value": [2.0, 1.0, 0.5]}},
         "sensor": {"type": "perspective", "to_world": mi.ScalarTransform4f().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}}}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img, _ = osc.render_path(sensor, seed=1, spp=8, max_depth=3)
    want = 0.5 / np.pi * np.array([2.0, 1.0, 0.5]) * 0.5
    assert np.allclose(img, want[None, None, :], rtol=2e-5)


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_product_host_shading_matches_oracle_with_a_directional_light(mi, O, H, mode, md):
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["sun"] = {"type": "directional", "direction": [0.3, -0.2, -1.0], "irradiance": {"type": "rgb", "value": [2.0, 1.5, 1.0]}}      # through the open front of the box
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((32, 32, 4), np.float32)
    assert H.hh_render(h, C.byref(sensor), mode, 4, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = 32; d2["sensor"]["film"]["height"] = 32
    osc2, s2 = O.scene_from_product(mi.load_dict(d2))
    ref2, _ = (osc2.render_path if mode == 0 else osc2.render_prb)(s2, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(ref2), O.develop(ref)) > 0.02
