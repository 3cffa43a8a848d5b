"""`orthographic` sensor (src/sensors/orthographic.cpp; HarSensor::projection = 1): the product's host lowering (har_orthographic_sensor: orthographic_projection of
sensor.h:272-307, inverted) with the oracle's and the product's ray generation, against
  * src/sensors/tests/test_orthographic.py:53-77 test02_sample_ray: ray origins on the near plane (dot(o, dir) = dot(origin + dir * near_clip, dir)), the centre sample
    runs along the camera direction (both origins x both directions of the reference's test, its 512 x 256 film, near 1 / far 35);
  * the geometry of orthographic_projection: a film of aspect a spans [-1, 1] x [-1 / a, 1 / a] of the camera's xy plane times the scale of to_world, crop windows select
    their part of it, maxt = far - near;
  * a render: the image of a diffuse plane under a directional light is CONSTANT (every ray meets the plane at the same angle), and equals the closed form;
  * each other: product host shading == oracle on a scene seen through an orthographic camera."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def camera(mi, o, d, width=512, height=256, scale=None, **film):
    T = mi.ScalarTransform4f().look_at(origin=o, target=list(np.asarray(o, float) + np.asarray(d, float)), up=[0, 1, 0])
    if scale is not None:
        T = T @ mi.ScalarTransform4f().scale(scale)
    return mi.load_dict({'type': 'orthographic', 'near_clip': 1.0, 'far_clip': 35.0, 'shutter_open': 1.5, 'shutter_close': 5, 'to_world': T,
                         'film': dict({'type': 'hdrfilm', 'width': width, 'height': height}, **film)})


def rays(O, sensor, pos):
    pos = np.asarray(pos, np.float32).reshape(-1, 2); n = pos.shape[0]
    s = O.Sensor(); C.memmove(C.byref(s), C.byref(sensor.har), C.sizeof(s))
    px = np.ascontiguousarray(pos[:, 0]); py = np.ascontiguousarray(pos[:, 1])
    o = np.zeros((3, n), np.float32); d = np.zeros((3, n), np.float32); mt = np.zeros(n, np.float32)
    O.lib().orc_sensor_sample_ray(C.byref(s), n, O.fp(px), O.fp(py), O.fp(o), O.fp(d), O.fp(mt))
    return o.T, d.T, mt


@pytest.mark.parametrize("origin", [[1.0, 0.0, 1.5], [1.0, 4.0, 1.5]])
@pytest.mark.parametrize("direction", [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
def test02_sample_ray(mi, O, origin, direction):
    """src/sensors/tests/test_orthographic.py:53-77"""
    cam = camera(mi, origin, direction)
    assert cam.har.projection == 1 and cam.near_clip == 1.0 and cam.far_clip == 35.0
    o, d, mt = rays(O, cam, [[0.2, 0.6], [0.1, 0.9], [0.2, 0.2]])
    want = np.dot(np.asarray(origin) + np.asarray(direction) * 1.0, direction)
    assert np.allclose(o @ np.asarray(direction), want)
    o, d, mt = rays(O, cam, [[0.5, 0.5]])
    assert np.allclose(d[0], direction, atol=1e-7) and np.allclose(mt, 34.0)
    # test03's finite differences: one pixel to the right / down moves the origin by 2 / width (the film spans [-1, 1]) along the camera's -x / -y, directions unchanged
    o2, d2, _ = rays(O, cam, [[0.5 + 1 / 512, 0.5], [0.5, 0.5 + 1 / 256]])
    assert np.allclose(d2, d[0][None, :], atol=1e-7)
    assert np.allclose(np.linalg.norm(o2 - o[0], axis=1), [2 / 512, 2 / 512 * 2 / 2], rtol=1e-4)        # aspect 2: y spans [-1/2, 1/2] over 256 pixels


def test_view_rectangle_scale_and_crop(mi, O):
    """orthographic_projection (sensor.h:272-307): sample (0, 0) is the top-left corner (+x, +y of the camera), the view is [-1, 1] x [-1/aspect, 1/aspect] times the
    scale in to_world; a crop window's samples address its part of the full film"""
    cam = camera(mi, [0, 0, 0], [0, 0, 1], width=200, height=100, scale=[10.0, 10.0, 1.0])
    o, d, _ = rays(O, cam, [[0, 0], [1, 1], [0.5, 0.5], [0.25, 0.5]])
    # look_at(origin, +z, up = y): camera x = world -x ... the corners are 10 (= scale) * (1, 1/2) away from the axis
    assert np.allclose(np.abs(o[0, :2]), [10.0, 5.0], rtol=1e-5) and np.allclose(o[0, :2], -o[1, :2], atol=1e-4)
    assert np.allclose(o[2, :2], 0.0, atol=1e-5) and np.allclose(o[:, 2], 1.0, atol=1e-5)          # on the near plane
    assert np.allclose(o[3, 0], o[0, 0] / 2, rtol=1e-5)
    assert np.allclose(d, [[0, 0, 1]] * 4, atol=1e-7)
    crop = camera(mi, [0, 0, 0], [0, 0, 1], width=200, height=100, scale=[10.0, 10.0, 1.0], crop_offset_x=100, crop_offset_y=50, crop_width=100, crop_height=50)
    oc, _, _ = rays(O, crop, [[0, 0], [1, 1]])
    assert np.allclose(oc[0, :2], 0.0, atol=1e-4) and np.allclose(oc[1, :2], o[1, :2], rtol=1e-5)    # the lower right quadrant


def plane_scene(mi, res=16):
    dirn = np.array([np.sin(np.pi / 3), 0.0, -np.cos(np.pi / 3)])
    return {"type": "scene", "rect": {"type": "rectangle", "to_world": mi.ScalarTransform4f().scale(10.0), "bsdf": {"type": "diffuse"}},
            "sun": {"type": "directional", "direction": [float(x) for x in dirn], "irradiance": {"type": "rgb", "value": [2.0, 1.0, 0.5]}},
            "sensor": {"type": "orthographic", "to_world": mi.ScalarTransform4f().look_at(origin=[0.3, -0.2, 4], target=[0.3, -0.2, 0], up=[0, 1, 0]) @ mi.ScalarTransform4f().scale([2.0, 2.0, 1.0]),
                       "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "box"}}}}


def test_constant_image_of_a_lit_plane(mi, O):
    scene = mi.load_dict(plane_scene(mi))
    osc, sensor = O.scene_from_product(scene)
    img, st = osc.render_path(sensor, seed=1, spp=4, max_depth=3)
    want = 0.5 / np.pi * np.array([2.0, 1.0, 0.5]) * 0.5
    assert np.allclose(img, want[None, None, :], rtol=2e-5)


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_product_host_shading_matches_oracle_through_an_orthographic_camera(mi, O, mode, md):
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); H.hh_scene_create.restype = C.c_void_p
    H.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p]; H.hh_scene_destroy.argtypes = [C.c_void_p]
    d = mi.cornell_box()
    film = dict(d["sensor"]["film"]); film["width"] = 32; film["height"] = 32
    d["sensor"] = {"type": "orthographic", "near_clip": 0.01, "far_clip": 100.0, "film": film,
                   "to_world": mi.ScalarTransform4f().look_at(origin=[0.1, 0.05, 3.9], target=[0, 0, 0], up=[0, 1, 0]) @ mi.ScalarTransform4f().scale([0.9, 0.9, 1.0])}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    assert sensor.projection == 1
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    f = np.zeros((32, 32, 4), np.float32)
    assert H.hh_render(h, C.byref(sensor), mode, 4, 8, md, 5, 0, 0, O.fp(f)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert np.abs(O.develop(ref)).max() > 0 and rel_l2(O.develop(f), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)


def test_orthographic_plugin_properties(mi):
    d = {'type': 'orthographic', 'fov': 45.0, 'film': {'type': 'hdrfilm', 'width': 8, 'height': 8}}
    with pytest.raises(RuntimeError, match="Unreferenced property"):
        mi.load_dict(d)
    # a scale in to_world is what sizes the view (orthographic.cpp:63-72 documents it); the perspective camera refuses one (perspective.cpp:143-146)
    mi.load_dict({'type': 'orthographic', 'to_world': mi.ScalarTransform4f().scale([10, 10, 1]), 'film': {'type': 'hdrfilm', 'width': 8, 'height': 8}})
    with pytest.raises(RuntimeError, match="Scale factors"):
        mi.load_dict({'type': 'perspective', 'to_world': mi.ScalarTransform4f().scale([10, 10, 1]), 'film': {'type': 'hdrfilm', 'width': 8, 'height': 8}})
