"""Edge cases of the render drivers and the array-valued calls, each against the oracle: scenes with no geometry / no emitters, depths -1 / 0 / 1, one sample, a 1 x 1 film and a
1 x 1 crop, Russian roulette from the first vertex, zero-length wavefronts, lane ranges that are empty or a single lane.  (What the reference's drivers accept, the variant must
accept: src/render/integrator.cpp:151-396, src/integrators/path.cpp:94-346.)"""
import numpy as np
import pytest

from tests.test_gpu_boundary import rel_l2

pytestmark = pytest.mark.gpu


def _cbox(mi, w, h, **film):
    d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = w; f["height"] = h; f.update(film)
    return d


def _both(mi, O, d, spp, seed, md, rr=5, kind="path"):
    d = dict(d); d["integrator"] = {"type": kind, "max_depth": md, "rr_depth": rr}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=seed).cpu().numpy()
    ref, st = (osc.render_path if kind == "path" else osc.render_prb)(sensor, seed=seed, spp=spp, max_depth=md, rr_depth=rr)
    return scene, img, ref, st


@pytest.mark.parametrize("md", [-1, 0, 1, 2])
def test_depth_limits(mi, O, md):
    """max_depth = -1 (no limit: Russian roulette ends the paths), 0 (nothing), 1 (directly visible emitters only), 2 (one bounce)"""
    for kind in ("path", "prb"):
        scene, img, ref, st = _both(mi, O, _cbox(mi, 24, 24), 8, 3, md, rr=3, kind=kind)
        assert img.shape == ref.shape and np.isfinite(img).all()
        if md == 0:
            assert not img.any() and not ref.any()
        else:
            assert rel_l2(img, ref) < 1e-4
        gst = scene.integrator().stats()
        assert gst["vertices"] == st.vertices and (md == 0 or gst["paths"] == st.paths), (md, kind, gst, st.paths, st.vertices)      # (max_depth = 0: nothing is launched, the path counter stays 0)


def test_tiny_films_and_single_samples(mi, O):
    for w, h, film, spp in ((1, 1, {}, 1), (1, 1, {}, 64), (7, 1, {}, 3), (1, 5, {"rfilter": {"type": "box"}}, 2),
                            (16, 16, {"crop_offset_x": 5, "crop_offset_y": 9, "crop_width": 1, "crop_height": 1}, 16),
                            (16, 16, {"crop_offset_x": 15, "crop_offset_y": 0, "crop_width": 1, "crop_height": 16, "sample_border": True}, 4)):
        scene, img, ref, st = _both(mi, O, _cbox(mi, w, h, **film), spp, 1, 5)
        assert img.shape == ref.shape and rel_l2(img, ref) < 1e-4, (w, h, film, spp)
        assert scene.integrator().stats()["vertices"] == st.vertices


def test_russian_roulette_from_the_first_vertex(mi, O):
    for rr in (1, 2):
        for kind in ("path", "prb"):
            scene, img, ref, st = _both(mi, O, _cbox(mi, 20, 20), 16, 2, 8, rr=rr, kind=kind)
            assert rel_l2(img, ref) < 1e-4 and scene.integrator().stats()["vertices"] == st.vertices, (rr, kind)


def test_scenes_without_geometry_or_without_light(mi, O):
    T = mi.ScalarTransform4f
    sensor = {"type": "perspective", "fov": 40, "to_world": T().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
              "film": {"type": "hdrfilm", "width": 12, "height": 10, "pixel_format": "rgb"}, "sampler": {"type": "independent", "sample_count": 4}}
    # only a sky: every ray escapes
    d = {"type": "scene", "sensor": sensor, "sky": {"type": "constant", "radiance": {"type": "rgb", "value": [0.3, 0.5, 0.8]}}}
    scene, img, ref, st = _both(mi, O, d, 4, 0, 4)
    assert np.allclose(img, [0.3, 0.5, 0.8], rtol=1e-5) and rel_l2(img, ref) < 1e-5
    # nothing at all: a black picture, no error
    d = {"type": "scene", "sensor": sensor}
    scene, img, ref, st = _both(mi, O, d, 4, 0, 4)
    assert not img.any() and not ref.any()
    # geometry, no emitter
    d = {"type": "scene", "sensor": sensor, "wall": {"type": "rectangle", "bsdf": {"type": "diffuse"}}}
    for kind in ("path", "prb"):
        scene, img, ref, st = _both(mi, O, d, 4, 0, 4, kind=kind)
        assert not img.any() and scene.integrator().stats()["vertices"] == st.vertices
    g = scene.integrator().render_backward(scene, None, np.ones((10, 12, 3), np.float32), seed=1, spp=4)
    assert all(not v.any() for v in g.values())


def test_zero_length_wavefronts_and_degenerate_lane_ranges(mi, O):
    import torch
    scene = mi.load_dict(_cbox(mi, 16, 16))
    empty = np.zeros((3, 0), np.float32)
    pi = scene.ray_intersect_preliminary(mi.Ray3f(empty, empty, np.zeros(0, np.float32)))
    assert pi.t.numel() == 0
    assert scene.ray_test(mi.Ray3f(empty, empty, np.zeros(0, np.float32))).numel() == 0
    integ = scene.integrator(); spp = 4
    full = integ.render_film(scene, seed=2, spp=spp)
    n = 16 * 16 * spp
    one = integ.render_film(scene, seed=2, spp=spp, lanes=(n // 2, n // 2 + 1))            # a single lane
    assert integ.stats()["paths"] == 1 and float(one[..., 3].sum()) > 0
    rest = integ.render_film(scene, seed=2, spp=spp, lanes=(0, n // 2)) + integ.render_film(scene, seed=2, spp=spp, lanes=(n // 2 + 1, n))
    assert torch.allclose(rest + one, full, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        integ.render_film(scene, seed=2, spp=spp, lanes=(n, n + 4))                         # past the end
