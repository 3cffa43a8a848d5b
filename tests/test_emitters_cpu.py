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


# ---------------------------------------------------------------- area lights on arbitrary triangle meshes (Mesh::sample_position)

def mesh_light_scene(mi, res=24, normals=True):
    """Cornell box lit by a small cube and by a tessellated, bumpy patch (PLY-like `mesh` with or without vertex normals) instead of the rectangle"""
    T = mi.ScalarTransform4f
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d.pop("light")
    d["cube_light"] = {"type": "cube", "to_world": T().translate([0.4, 0.55, 0.1]).rotate([0, 1, 0], 30).scale([0.12, 0.05, 0.2]),
                       "bsdf": {"type": "ref", "id": "white"}, "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [9.0, 7.0, 4.0]}}}
    n = 5
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    P = np.stack([0.25 * xs.ravel() - 0.45, 0.9 - 0.06 * np.cos(2 * xs.ravel()) * np.cos(2 * ys.ravel()), 0.25 * ys.ravel()], 1).astype(np.float32)
    F = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i; F += [[a, a + 1, a + n], [a + 1, a + n + 1, a + n]]          # faces downwards (-y)
    mesh = {"type": "mesh", "positions": P, "faces": np.asarray(F, np.uint32), "bsdf": {"type": "ref", "id": "white"},
            "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [4.0, 6.0, 8.0]}}}
    if normals:
        nrm = np.stack([0.3 * np.sin(2 * xs.ravel()), -np.ones(n * n), 0.3 * np.sin(2 * ys.ravel())], 1); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        mesh["normals"] = nrm.astype(np.float32)
    d["patch_light"] = mesh
    return d


def test_mesh_emitter_sampling_statistics(mi, O):
    """the direct illumination of a diffuse floor by a mesh light, estimated by emitter sampling + MIS (max_depth = 2) and by BSDF sampling
    alone (the same estimator with the light's sampling probability ignored is not available; instead: two independent seeds and the oracle's
    energy balance): the image is the same for the two light-sampling-heavy and bsdf-heavy regimes, i.e. MIS weights sum to one"""
    from tests.test_cpu_host import oracle_scene_from
    d = mesh_light_scene(mi, 16)
    scene = mi.load_dict(d)
    types = sorted(e["type"] for e in scene.emitters)
    assert types == [3, 3]
    osc, sensor = oracle_scene_from(O, scene)
    a, _ = osc.render_path(sensor, seed=1, spp=512, max_depth=3)
    b, _ = osc.render_path(sensor, seed=2, spp=512, max_depth=3)
    assert a.mean() > 0.02 and abs(a.mean() / b.mean() - 1) < 0.03
    # uniform area density: the oracle's mesh sampler visits faces proportionally to their area (Mesh::build_pmf)
    V = scene.meshes[[m["key"] for m in scene.meshes].index("patch_light")]
    P, Fi = V["V"][:, :3], V["F"][:, :3]
    areas = 0.5 * np.linalg.norm(np.cross(P[Fi[:, 1]] - P[Fi[:, 0]], P[Fi[:, 2]] - P[Fi[:, 0]]), axis=1)
    em = [e for e in scene.emitters if e["mesh"] == [m["key"] for m in scene.meshes].index("patch_light")][0]
    assert em["type"] == 3 and areas.sum() > 0


@pytest.mark.parametrize("mode,normals", [(0, True), (1, True), (0, False)])
def test_mesh_emitter_product_shading_matches_oracle(mi, O, mode, normals):
    """host-compiled shade_lane (extended-emitter variant) vs the oracle with mesh area lights (with / without vertex normals)"""
    import ctypes as C
    from tests.test_cpu_host import oracle_scene_from, _harness_scene, rel_l2
    from tests.test_envmap_cpu import H as _  # noqa: F401  (fixture module import keeps pytest's collection order stable)
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Hh = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    Hh.hh_scene_create.restype = C.c_void_p; Hh.hh_scene_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    Hh.hh_scene_destroy.argtypes = [C.c_void_p]
    Hh.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    scene = mi.load_dict(mesh_light_scene(mi, 24, normals))
    osc, sensor = oracle_scene_from(O, scene)
    h = _harness_scene(Hh, scene)
    film = np.zeros((24, 24, 4), np.float32)
    assert Hh.hh_render(h, C.byref(sensor), mode, 3, 8, 6, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=8, max_depth=6, raw=True, threads=2)
    assert O.develop(ref).mean() > 0.02
    assert rel_l2(O.develop(film), O.develop(ref)) < 2e-5
    Hh.hh_scene_destroy(h)


def test_mesh_emitter_equals_rectangle_light(mi, O):
    """the Cornell box light as a two-triangle `mesh` (Mesh::sample_position: face pmf + uniform triangle) vs the analytic `rectangle`
    (Rectangle::sample_position): different sample mappings of the same area density -> statistically identical images (Z-test)"""
    from tests import ztest
    from tests.test_cpu_host import oracle_scene_from
    res = 20
    def scene_dict(as_mesh):
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
        d["sensor"]["film"]["rfilter"] = {"type": "box"}
        if as_mesh:
            rect = mi.load_dict({"type": "scene", "white": d["white"], "r": d["light"]}).meshes[0]       # the baked rectangle: 4 vertices, 2 faces
            light = d.pop("light")
            d["light"] = {"type": "mesh", "positions": rect["V"][:, :3], "normals": rect["V"][:, 3:6], "texcoords": rect["V"][:, 6:8],
                          "faces": rect["F"][:, :3], "bsdf": light["bsdf"], "emitter": light["emitter"]}
        return d
    sa = mi.load_dict(scene_dict(False)); sb = mi.load_dict(scene_dict(True))
    assert sa.emitters[0]["type"] == 0 and sb.emitters[0]["type"] == 3
    osa, sensor = oracle_scene_from(O, sa); osb, _ = oracle_scene_from(O, sb)
    ref_mean, ref_var, n_ref = ztest.oracle_reference(osa, sensor, spp_b=8, batches=96, max_depth=4)
    img, _ = osb.render_path(sensor, seed=99, spp=512, max_depth=4)
    ok, pmin, alpha = ztest.accept(img, 512, ref_mean, ref_var, n_ref)
    assert ok, (pmin, alpha)


def hide_emitters_scene(mi, res=32):
    """the Cornell box seen by its own camera, with the back wall replaced by a constant sky: the ceiling light and the sky are in view"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d.pop("back"); d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.2, 0.3, 0.5]}}
    return d


def test_oracle_hide_emitters(mi, O):
    """Integrator property `hide_emitters` (path.cpp:114-115,177-190; prb.py:112-118,146-148): camera rays pass through area emitters and
    do not see the environment; everything the emitters light stays lit"""
    from tests.test_cpu_host import oracle_scene_from
    scene = mi.load_dict(hide_emitters_scene(mi))
    osc, sensor = oracle_scene_from(O, scene)
    kw = dict(seed=1, spp=16, max_depth=4)
    shown, _ = osc.render_path(sensor, **kw)
    osc.set_hide_emitters(True)
    hidden, _ = osc.render_path(sensor, **kw)
    hidden_prb, _ = osc.render_prb(sensor, **kw)
    osc.set_hide_emitters(False)
    light = shown[:, :, 0] > 6.0
    assert light.sum() >= 4 and hidden[light].max() < 4.0             # the light is in view ... and gone: those pixels show the ceiling behind it
    sky = np.abs(shown - np.array([0.2, 0.3, 0.5], np.float32)).max(axis=2) < 1e-4
    assert sky.sum() >= 16 and np.abs(hidden[sky]).max() < 0.05         # camera rays that escape see nothing (a little filter bleed from the walls)
    # pixels whose 5 x 5 filter footprint holds neither: identical, sample for sample
    same = np.abs(shown - hidden).max(axis=2) <= 1e-5 * np.maximum(np.abs(shown).max(axis=2), 1e-3)
    assert same.mean() > 0.25
    # prb: same estimator in expectation
    assert abs(hidden_prb.mean() / hidden.mean() - 1) < 0.1


def test_oracle_alpha_channel(mi, O):
    """alpha of `rgba` films = filtered valid-sample mask: PathIntegrator counts a visible environment as valid (path.cpp:114-115), prb only
    samples that met a surface (prb.py:332), hide_emitters takes the environment (and emitters in front of it) out"""
    from tests.test_cpu_host import oracle_scene_from
    scene = mi.load_dict(hide_emitters_scene(mi))            # no back wall: sky behind
    osc, sensor = oracle_scene_from(O, scene)
    kw = dict(seed=1, spp=8, max_depth=4)
    osc.set_alpha_only(True)
    a_path, _ = osc.render_path(sensor, **kw); a_prb, _ = osc.render_prb(sensor, **kw)
    osc.set_hide_emitters(True)
    a_hidden, _ = osc.render_path(sensor, **kw)
    osc.set_hide_emitters(False); osc.set_alpha_only(False)
    assert np.abs(a_path - 1.0).max() < 1e-5                                  # sky or surface everywhere
    assert a_prb.min() < 0.05 and a_prb.max() > 0.999                         # the sky pixels are holes for prb
    assert np.abs(a_prb - a_hidden).max() < 1e-5                              # ... and for path with hide_emitters (the light hides nothing valid here)
    assert (a_path[..., 0] == a_path[..., 1]).all()
