"""CPU tests (no GPU): oracle known-answer tests, C ABI surface, host lowering, and the
product's HAR_HD logic (BVH8 build + traversal, shading stages, film) executed through the
host test harness against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def oracle_scene_from(O, scene):
    """moved next to the oracle (oracle/oracle.py: scene_from_product); kept here for the tests that import it from this module"""
    return O.scene_from_product(scene)


# ---------------------------------------------------------------- oracle KATs

def test_tea_kats(O):
    """src/core/tests/test_random.py:8-28"""
    L = O.lib()
    f32 = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214, (1, 4): 0.008385419845581055,
           (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013, (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    f64 = {(1, 1): 0.5424730799533735, (1, 2): 0.5079905082233922, (1, 3): 0.4171962610608142, (1, 4): 0.008385529523330604,
           (1, 5): 0.80855288317879, (2, 1): 0.6939880404156831, (3, 1): 0.6978365636630994, (4, 1): 0.48973647949223253}
    for (a, b), v in f32.items():
        assert L.orc_sample_tea_float32(a, b, 4) == np.float32(v)
    for (a, b), v in f64.items():
        assert L.orc_sample_tea_float64(a, b, 4) == v


def test_pcg32_published_vector(O):
    """pcg32-demo (pcg-random.org), seed(42, 54): the published first six outputs."""
    L = O.lib(); si = (C.c_uint64 * 2)()
    L.orc_pcg32_seed(42, 54, si)
    assert [L.orc_pcg32_next_uint32(si) for _ in range(6)] == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_diffuse_closed_form(O):
    """src/bsdfs/tests/test_diffuse.py:16-39"""
    L = O.lib(); refl = O.f32([0.5, 0.5, 0.5]); wi = O.f32([0, 0, 1])
    for i in range(20):
        th = i / 19.0 * (np.pi / 2); wo = O.f32([np.sin(th), 0, np.cos(th)])
        val = np.empty(3, np.float32); pdf = C.c_float()
        L.orc_diffuse_eval_pdf(O.fp(refl), O.fp(wi), O.fp(wo), O.fp(val), C.byref(pdf))
        assert np.isclose(pdf.value, max(wo[2], 0) / np.pi, atol=1e-7) and np.isclose(val[0], 0.5 * max(wo[2], 0) / np.pi, atol=1e-7)


def test_cosine_hemisphere_and_sincos(O):
    L = O.lib(); rng = np.random.default_rng(0)
    for _ in range(2000):
        s = rng.random(2).astype(np.float32); out = np.empty(3, np.float32)
        L.orc_square_to_cosine_hemisphere(O.fp(s), O.fp(out))
        assert abs(np.linalg.norm(out) - 1) < 1e-5 and out[2] >= 0
    for x in np.linspace(-3, 3, 200):
        c = C.c_float(); s = L.orc_sincos(float(x), C.byref(c))
        assert abs(s - np.sin(np.float32(x))) < 3e-7 and abs(c.value - np.cos(np.float32(x))) < 3e-7


def test_cornell_pixel_kat(O):
    """src/integrators/tests/test_integrators.py:28-53"""
    sd, sensor = O.cornell_box(256, 256, crop=(124, 36, 1, 1))
    img, _ = O.OracleScene(sd).render_path(sensor, spp=64, max_depth=1)
    assert np.allclose(img.reshape(3), [18.387, 13.9873, 6.75357], rtol=1e-5)


def _stairs(O, n_steps=20):
    """src/render/tests/test_kdtrees.py:8-36"""
    v = np.zeros((4 * n_steps, 3), np.float32); f = []
    for i in range(n_steps):
        h = i / n_steps; s1 = i / n_steps; s2 = (i + 1) / n_steps; k = 4 * i
        v[k] = [0, s1, h]; v[k + 1] = [1, s1, h]; v[k + 2] = [0, s2, h]; v[k + 3] = [1, s2, h]
        f += [[k, k + 1, k + 2], [k + 1, k + 3, k + 2]]
        if i < n_steps - 1:
            f += [[k + 2, k + 3, k + 5], [k + 5, k + 4, k + 2]]
    V = np.zeros((v.shape[0], 8), np.float32); V[:, :3] = v
    F = np.zeros((len(f), 4), np.uint32); F[:, :3] = np.asarray(f, np.uint32)
    sd = O.SceneData(); sd.bsdfs = [(0, -1, [0.5, 0.5, 0.5])]; sd.add_mesh(V, F, 0, -1, 0); sd.top_mesh_count = 1
    return sd


def test_stairs_analytic_depth(O):
    """test_kdtrees.py:50-81: accel == brute force == analytic t = 2 - floor(y*20)/20"""
    sd = _stairs(O); osc = O.OracleScene(sd)
    n = 128; inv = 1.0 / (n - 1)
    xs, ys = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    o = np.stack([xs.ravel() * inv, ys.ravel() * inv, np.full(xs.size, 2.0)]).astype(np.float32)
    d = np.tile(np.array([[0], [0], [-1]], np.float32), (1, o.shape[1]))
    maxt = np.full(o.shape[1], 100, np.float32)
    a = osc.ray_intersect(o, d, maxt, naive=True); b = osc.ray_intersect(o, d, maxt, naive=False)
    expected = 2.0 - np.floor((ys.ravel() * inv) * 20) / 20
    assert np.allclose(a[0], expected, atol=1e-6) and np.array_equal(a[0], b[0])
    assert osc.ray_test(o, d, maxt).all()


def test_film_put_vs_numpy(O):
    """src/render/tests/test_imageblock.py:37-124 (coalesced gaussian put vs an independent NumPy splat)."""
    sd, sensor = O.cornell_box(32, 24)
    rng = np.random.default_rng(3); n = 500
    px = rng.uniform(-1, 33, n).astype(np.float32); py = rng.uniform(-1, 25, n).astype(np.float32)
    vals = rng.random((n, 4)).astype(np.float32)
    film = np.zeros((24, 32, 4), np.float32)
    O.lib().orc_film_put(C.byref(sensor), n, O.fp(px), O.fp(py), O.fp(vals), O.fp(film))
    ref = np.zeros((24, 32, 4), np.float64)
    w = lambda x: max(float(O.lib().orc_rfilter_eval(1, 0.5, float(x))), 0.0)
    for i in range(n):
        x0 = int(np.floor(px[i])) - 2; y0 = int(np.floor(py[i])) - 2
        for ys in range(5):
            for xs in range(5):
                x, y = x0 + xs, y0 + ys
                if 0 <= x < 32 and 0 <= y < 24:
                    ref[y, x] += vals[i] * w(np.float32(x0 + .5) - px[i] + xs) * w(np.float32(y0 + .5) - py[i] + ys)
    assert np.allclose(film, ref, atol=1e-4)
    assert abs(O.lib().orc_rfilter_eval(1, 0.5, 0.0) - 0.99925) < 1e-4 and abs(O.lib().orc_rfilter_eval(1, 0.5, 2.0)) < 1e-5


def test_oracle_ad_linearity(O):
    """src/render/tests/test_ad.py:55-134 adapted to an area light: with max_depth=2 the image is linear
    in each albedo, so loss(rho + lr) == loss(rho) + lr * dloss/drho for the same seed."""
    sd, sensor = O.cornell_box(16, 16)
    osc = O.OracleScene(sd)
    seed, spp = 3, 32
    img1, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=2)
    grad_in = np.ones((16, 16, 3), np.float32)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=seed, spp=spp, max_depth=2)
    lr = 0.01
    rho = np.array(O.CBOX_WHITE, np.float32); rho2 = rho.copy(); rho2[0] += lr
    osc.set_reflectance(0, rho2)
    img2, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=2)
    assert np.isclose(img2.sum(), img1.sum() + lr * g_refl[0, 0], rtol=2e-4)


def test_oracle_prb_gradient_vs_finite_differences(O):
    """test_ad_integrators.py:1463-1511 recipe at test size: central differences of the oracle's own prb primal."""
    sd, sensor = O.cornell_box(12, 12)
    osc = O.OracleScene(sd)
    seed, spp, md = 5, 64, 4
    grad_in = np.ones((12, 12, 3), np.float32)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=seed, spp=spp, max_depth=md)
    eps = 1e-2
    rho = np.array(O.CBOX_WHITE, np.float32)
    a = rho.copy(); a[1] += eps; osc.set_reflectance(0, a); ip, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=md)
    b = rho.copy(); b[1] -= eps; osc.set_reflectance(0, b); im, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=md)
    fd = (ip.astype(np.float64).sum() - im.astype(np.float64).sum()) / (2 * eps)
    assert abs(fd - g_refl[0, 1]) / abs(fd) < 2e-2


# ---------------------------------------------------------------- C ABI surface

def test_capi_exports_every_declared_symbol(mi):
    hdr = open(os.path.join(ROOT, "include", "hip_ad_rgb.h")).read()
    declared = set(re.findall(r"\b(har_[a-z0-9_]+)\s*\(", hdr))
    L = C.CDLL(mi.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    from mitsuba3_amd import _capi
    assert declared == set(_capi.SIGNATURES)


def test_error_behaviour_without_side_effects(mi):
    with pytest.raises(ImportError):
        mi.set_variant("cuda_ad_rgb")
    with pytest.raises(RuntimeError):
        mi.load_dict({"type": "principled"})                      # a plugin outside the scope of hip_ad_rgb
    with pytest.raises(RuntimeError):
        mi.load_dict({"type": "dielectric", "int_ior": -0.5})       # src/bsdfs/tests/test_dielectric.py:24-26
    with pytest.raises(RuntimeError):
        mi.load_dict({"type": "twosided", "b": {"type": "dielectric"}})   # twosided.cpp:79-83
    with pytest.raises(RuntimeError):
        mi.load_dict({"type": "path", "rr_depth": 0})
    with pytest.raises(RuntimeError):
        mi.load_dict({"type": "path", "max_depth": -2})
    h = C.c_void_p()
    assert mi.lib().har_integrator_create(7, 8, 5, 0, C.byref(h)) != 0 and b"unknown" in mi.lib().har_last_error()


# ---------------------------------------------------------------- host lowering vs oracle builders

def test_host_lowering_bit_identical_to_oracle(mi, O):
    scene = mi.load_dict(mi.cornell_box())
    sd, osens = O.cornell_box(256, 256)
    assert len(scene.meshes) == len(sd.meshes) == 8
    for a, b in zip(scene.meshes, sd.meshes):
        assert np.array_equal(a["V"], b["V"]) and np.array_equal(a["F"], b["F"]) and a["bsdf"] == b["bsdf"] and a["emitter"] == b["emitter"]
    s = scene.sensors()[0].har
    assert bytes(s) == bytes(osens)
    e0, e1 = scene.emitters[0], sd.emitters[0]
    assert np.array_equal(e0["to_world"], e1["to_world"]) and np.array_equal(e0["normal"], e1["normal"]) and e0["inv_area"] == e1["inv_area"]


# ---------------------------------------------------------------- product logic through the host harness

@pytest.fixture(scope="module")
def H(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p
    L.hh_scene_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.hh_scene_destroy.argtypes = [C.c_void_p]
    L.hh_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.hh_trace.argtypes = [C.c_void_p, C.c_uint32] + [O.c_f32p] * 3 + [C.c_int, C.c_int] + [O.c_f32p] * 3 + [O.c_u32p] * 3 + [C.POINTER(C.c_uint8)]
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    return L


def _harness_scene(H, scene):
    d = scene.desc(); err = C.create_string_buffer(256)
    h = C.c_void_p(H.hh_scene_create(C.byref(d), err, 256))
    assert h, err.value
    return h


def _trace_both(H, O, scene, osc, n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-1, 1, (3, n)).astype(np.float32); d = rng.normal(size=(3, n)).astype(np.float32); d /= np.linalg.norm(d, axis=0)
    d = np.ascontiguousarray(d, np.float32); maxt = np.full(n, 3.402823466e+38, np.float32)
    ref = osc.ray_intersect(o, d, maxt, naive=True)
    h = _harness_scene(H, scene)
    t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
    p = np.empty(n, np.uint32); s = np.empty(n, np.uint32); i = np.empty(n, np.uint32); hf = np.empty(n, np.uint8)
    hfp = hf.ctypes.data_as(C.POINTER(C.c_uint8))
    for naive in (0, 1, 2, 3):      # reference loop, brute force, the persistent kernels' resumable traversal (2: the instantiation the launchers pick -- FLAT without a TLAS; 3: generic)
        assert H.hh_trace(h, n, O.fp(o), O.fp(d), O.fp(maxt), naive, 0, O.fp(t), O.fp(u), O.fp(v), O.up(p), O.up(s), O.up(i), hfp) == 0
        hit = np.isfinite(ref[0])
        assert np.array_equal(t, ref[0]) and np.array_equal(u[hit], ref[1][hit]) and np.array_equal(v[hit], ref[2][hit])
        assert np.array_equal(p[hit], ref[3][hit]) and np.array_equal(s[hit], ref[4][hit]) and np.array_equal(i[hit], ref[5][hit])
    maxt2 = rng.uniform(0.05, 2.5, n).astype(np.float32)
    occluded = osc.ray_test(o, d, maxt2)
    for naive in (0, 2, 3):
        H.hh_trace(h, n, O.fp(o), O.fp(d), O.fp(maxt2), naive, 1, O.fp(t), O.fp(u), O.fp(v), O.up(p), O.up(s), O.up(i), hfp)
        assert np.array_equal(hf.astype(bool), occluded)
    H.hh_scene_destroy(h)


def test_bvh8_traversal_equals_brute_force_cornell(mi, O, H):
    scene = mi.load_dict(mi.cornell_box())
    osc, _ = oracle_scene_from(O, scene)
    _trace_both(H, O, scene, osc, 100000, 1)


def test_bvh8_two_level_instanced(mi, O, H):
    scene = mi.load_dict(mi.instanced_spheres_scene(width=16, height=16, spp=1, grid=3, n_u=16, n_v=8))
    osc, _ = oracle_scene_from(O, scene)
    _trace_both(H, O, scene, osc, 60000, 2)


def test_bvh8_flattened_many_triangles(mi, O, H):
    scene = mi.load_dict(mi.instanced_spheres_scene(width=16, height=16, spp=1, grid=4, n_u=24, n_v=12, flatten=True))
    osc, _ = oracle_scene_from(O, scene)
    h = _harness_scene(H, scene); info = (C.c_uint64 * 4)(); H.hh_scene_info(h, info); H.hh_scene_destroy(h)
    assert info[1] == 16 * 2 * 24 * 12 + 12 and info[2] <= 16
    _trace_both(H, O, scene, osc, 40000, 3)


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_shading_stages_match_oracle(mi, O, H, mode, md):
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 40; d["sensor"]["film"]["height"] = 40
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    h = _harness_scene(H, scene)
    film = np.zeros((40, 40, 4), np.float32)
    assert H.hh_render(h, C.byref(sensor), mode, 4, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=4, spp=8, max_depth=md, raw=True, threads=2)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)


def test_shading_textured_and_instanced(mi, O, H):
    d = mi.textured_cornell_box(res=24, tex_res=16, spp=4)
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    h = _harness_scene(H, scene); film = np.zeros((24, 24, 4), np.float32)
    H.hh_render(h, C.byref(sensor), 1, 0, 4, 6, 5, 0, 0, O.fp(film))
    ref, _ = osc.render_prb(sensor, seed=0, spp=4, max_depth=6, raw=True)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)
    scene = mi.load_dict(mi.instanced_spheres_scene(width=24, height=24, spp=4, grid=3, n_u=12, n_v=6))
    osc, sensor = oracle_scene_from(O, scene)
    h = _harness_scene(H, scene); film = np.zeros((24, 24, 4), np.float32)
    H.hh_render(h, C.byref(sensor), 0, 1, 4, 8, 5, 0, 0, O.fp(film))
    ref, _ = osc.render_path(sensor, seed=1, spp=4, max_depth=8, raw=True)
    assert rel_l2(O.develop(film), O.develop(ref)) < 1e-6
    H.hh_scene_destroy(h)


# ------------------------------------------------------------------ the benchmark scenes: product lowering == the oracle's own lowering

@pytest.mark.parametrize("flatten", [False, True], ids=["instanced", "flattened"])
def test_benchmark_scene_lowering_matches_the_oracles_own(mi, O, flatten):
    """`mi.load_dict(mi.instanced_spheres_scene(...))` (product host code: ScalarTransform4f chain, mesh baking / instance matrices, sensor) against
    `O.benchmark_spheres_scene(...)` (the oracle's transform / baking / sensor code): same baked vertices, same instance matrices, same sensor
    record -- the GPU headline tests feed the oracle from the latter, this test says on the CPU where a difference would come from"""
    import ctypes as C
    res = 48
    d = mi.instanced_spheres_scene(width=res, height=res, spp=4, grid=4, n_u=20, n_v=10, flatten=flatten, textured=True, tex_res=16)
    scene = mi.load_dict(d)
    sd, sensor = O.benchmark_spheres_scene(res, res, grid=4, n_u=20, n_v=10, flatten=flatten, textured=True, tex_res=16)
    assert len(scene.meshes) == len(sd.meshes) and scene.top_mesh_count == sd.top_mesh_count
    for a, b in zip(scene.meshes, sd.meshes):
        assert np.array_equal(a["F"], b["F"]) and a["bsdf"] == b["bsdf"] and a["emitter"] == b["emitter"]
        assert np.abs(a["V"] - b["V"]).max() <= 1e-6
    assert len(scene.instances) == len(sd.instances)
    for (g, tw, to), (g2, tw2, to2) in zip(scene.instances, sd.instances):
        assert g == g2 and np.abs(np.asarray(tw, np.float32) - tw2).max() <= 1e-6 and np.abs(np.asarray(to, np.float32) - to2).max() <= 1e-5
    mine = scene.sensors()[0].har
    for name, _ in sensor._fields_:
        a, b = getattr(mine, name), getattr(sensor, name)
        a = np.asarray(list(a) if hasattr(a, "__len__") else a, np.float64); b = np.asarray(list(b) if hasattr(b, "__len__") else b, np.float64)
        assert np.allclose(a, b, rtol=1e-6, atol=1e-7), name
    assert np.array_equal(scene.textures[0], sd.textures[0])


def test_materials_scene_lowering_matches_the_oracles_own(mi, O):
    """the `materials=True` variant (rough plastic walls; twosided GGX conductor, diffuse and glass spheres): the BSDF records the product parses from the
    dict against the records `O.benchmark_spheres_scene(materials=True)` writes down from the plugins' documented defaults (distribution, visible
    normals, named IORs, eta in single precision, slot defaults), and the two oracle scenes render the same paths"""
    res, spp = 40, 4
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=16, n_v=8, flatten=True, materials=True)
    scene = mi.load_dict(d)
    sd, sensor = O.benchmark_spheres_scene(res, res, grid=3, n_u=16, n_v=8, flatten=True, materials=True)
    types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3, "conductor": 4, "plastic": 5}
    assert len(scene.bsdf_objs) == len(sd.bsdfs) == 4
    for b, rec in zip(scene.bsdf_objs, sd.bsdfs):
        x = rec[3] if len(rec) > 3 else {}
        assert types[b.kind] == rec[0] and rec[1] == -1 and np.array_equal(np.asarray(b.value, np.float32), np.asarray(rec[2], np.float32)) and b.flags == x.get("flags", 0), (b.kind, rec)
        if b.kind != "diffuse":
            assert np.array_equal(np.asarray(b.value2, np.float32), np.asarray(x.get("reflectance2", (0, 0, 0)), np.float32))
            assert np.float32(b.eta) == np.float32(x.get("eta", 1.0)) and np.float32(b.alpha_u) == np.float32(x.get("alpha_u", 0.1)) and np.float32(b.alpha_v) == np.float32(x.get("alpha_v", 0.1))
            assert np.array_equal(np.asarray(b.eta_c, np.float32), np.asarray(x.get("eta_c", (0, 0, 0)), np.float32)) and np.array_equal(np.asarray(b.k_c, np.float32), np.asarray(x.get("k_c", (1, 1, 1)), np.float32))
    assert [m["bsdf"] for m in scene.meshes] == [m["bsdf"] for m in sd.meshes]
    oa, sa = O.scene_from_product(scene)
    ia, sta = oa.render_path(sa, seed=0, spp=spp, max_depth=8)
    ib, stb = O.OracleScene(sd).render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert sta.vertices == stb.vertices and np.abs(ia - ib).max() <= 1e-5 * np.abs(ia).max()


# ------------------------------------------------------------------ Film::sample_border (film.cpp:29-32, integrator.cpp:162-165, 322-339)

@pytest.mark.parametrize("rf,crop", [("gaussian", None), ("gaussian", (5, 3, 20, 17)), ("tent", None), ("box", None)], ids=["gaussian", "gaussian-crop", "tent", "box"])
def test_sample_border_host_pipeline_matches_oracle(mi, O, H, rf, crop):
    """`sample_border`: the lane -> pixel map covers crop_size + 2 * border_size pixels, shifted back by the border; the film keeps the crop size and
    border samples are clipped.  Product (host compilation of the kernels' headers) vs oracle, plus what the property means: more samples
    (paths = (W + 2b)(H + 2b) spp), larger filter-weight sums along the edges, nothing changes for the border-free box filter."""
    res, spp = 28, 4
    d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = res; f["height"] = res; f["rfilter"] = {"type": rf}; f["sample_border"] = True
    if crop:
        f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = crop
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    assert sensor.sample_border == 1
    w, h = (crop[2], crop[3]) if crop else (res, res)
    b = {"gaussian": 2, "tent": 1, "box": 0}[rf]
    assert scene.sensors()[0].film().sample_grid() == (w + 2 * b, h + 2 * b)
    hnd = _harness_scene(H, scene)
    film = np.zeros((h, w, 4), np.float32)
    assert H.hh_render(hnd, C.byref(sensor), 0, 2, spp, 8, 5, 0, 0, O.fp(film)) == 0
    ref, st = osc.render_path(sensor, seed=2, spp=spp, max_depth=8, raw=True, threads=2)
    assert st.paths == (w + 2 * b) * (h + 2 * b) * spp
    assert rel_l2(film, ref) < 1e-6
    H.hh_scene_destroy(hnd)
    # against the same film without the property: identical for the box filter, heavier edges otherwise
    sensor.sample_border = 0
    plain, st0 = osc.render_path(sensor, seed=2, spp=spp, max_depth=8, raw=True, threads=2)
    assert st0.paths == w * h * spp
    if b == 0:
        assert np.array_equal(plain, ref)
    else:
        assert ref[0, :, 3].sum() > 1.03 * plain[0, :, 3].sum() and ref[:, -1, 3].sum() > 1.03 * plain[:, -1, 3].sum()
        inner = (slice(2 * b + 1, h - 2 * b - 1), slice(2 * b + 1, w - 2 * b - 1))
        assert abs(ref[inner][..., 3].mean() / plain[inner][..., 3].mean() - 1) < 0.05      # interior weights: same density of samples


def test_principal_point_offset_host_pipeline_and_meaning(mi, O, H):
    """PerspectiveCamera `principal_point_offset_x / _y` (src/sensors/perspective.cpp:147-150, 213-221): sample_ray adds film_size * offset / crop_size to the film
    position.  Product (host build of the kernels' headers) vs oracle on a render; and what the property means: an offset of k / film_width is the picture of
    the crop window moved by k pixels"""
    res, spp = 32, 4
    def scene_with(ppo, crop):
        d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = res; f["height"] = res; f["rfilter"] = {"type": "box"}
        f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = crop
        d["sensor"]["principal_point_offset_x"], d["sensor"]["principal_point_offset_y"] = ppo
        return mi.load_dict(d)
    scene = scene_with((3 / res, -2 / res), (6, 8, 16, 12))
    osc, sensor = oracle_scene_from(O, scene)
    assert np.isclose(sensor.principal_point_offset_x, 3 / res) and np.isclose(sensor.principal_point_offset_y, -2 / res)
    hnd = _harness_scene(H, scene)
    film = np.zeros((12, 16, 4), np.float32)
    assert H.hh_render(hnd, C.byref(sensor), 0, 5, spp, 8, 5, 0, 0, O.fp(film)) == 0
    ref, _ = osc.render_path(sensor, seed=5, spp=spp, max_depth=8, raw=True, threads=2)
    assert rel_l2(film, ref) < 1e-6
    H.hh_scene_destroy(hnd)
    # the same rays as the window moved by (+3, -2) pixels without an offset
    n = 4000
    p = np.random.default_rng(3).uniform(0, 1, (2, n)).astype(np.float32)
    moved = scene_with((0.0, 0.0), (9, 6, 16, 12))
    _, s2 = oracle_scene_from(O, moved)
    rays = []
    for sn in (sensor, s2):
        o = np.empty((3, n), np.float32); dd = np.empty((3, n), np.float32); mt = np.empty(n, np.float32)
        O.lib().orc_sensor_sample_ray(C.byref(sn), n, O.fp(np.ascontiguousarray(p[0])), O.fp(np.ascontiguousarray(p[1])), O.fp(o), O.fp(dd), O.fp(mt))
        rays.append((o, dd, mt))
    assert np.allclose(rays[0][1], rays[1][1], atol=2e-6) and np.allclose(rays[0][0], rays[1][0], atol=2e-6) and np.allclose(rays[0][2], rays[1][2], rtol=1e-5)
    plain = scene_with((0.0, 0.0), (6, 8, 16, 12))
    _, s3 = oracle_scene_from(O, plain)
    o3 = np.empty((3, n), np.float32); d3 = np.empty((3, n), np.float32); m3 = np.empty(n, np.float32)
    O.lib().orc_sensor_sample_ray(C.byref(s3), n, O.fp(np.ascontiguousarray(p[0])), O.fp(np.ascontiguousarray(p[1])), O.fp(o3), O.fp(d3), O.fp(m3))
    assert np.abs(d3 - rays[0][1]).max() > 1e-2                      # the offset does something


def test_elementary_functions_accuracy_and_host_device_agreement(O):
    """dr::exp / log / erf / atan2 / acos / tan (Dr.Jit, NOT IN TREE): the oracle's Cephes-style restatements (orc_math.h) against double
    precision, and the product's own versions (har_math.h, compiled for the host) against the oracle's BIT FOR BIT -- they are written
    independently but must describe the same arithmetic, which is what makes paths through rough BSDFs identical on the device."""
    import math
    L = O.lib()
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_math_fn.restype = C.c_float; H.hh_math_fn.argtypes = [C.c_int, C.c_float, C.c_float]
    rng = np.random.default_rng(7)
    n = 20000
    cases = {
        0: (rng.uniform(-86, 88, n), None, math.exp, 1.5),
        1: (np.exp(rng.uniform(-87, 88, n)), None, math.log, 1.5),
        2: (rng.uniform(-4.5, 4.5, n), None, math.erf, 1.5),
        3: (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), math.atan2, 4.0),
        4: (rng.uniform(-1, 1, n), None, math.acos, 2.0),
        5: (rng.uniform(-1.5, 1.5, n), None, math.tan, 3.0),
    }
    for fn, (xs, ys, ref, ulps) in cases.items():
        xs = xs.astype(np.float32); ys = np.zeros_like(xs) if ys is None else ys.astype(np.float32)
        worst = 0.0
        for x, y in zip(xs, ys):
            a = L.orc_math_fn(fn, float(x), float(y)); b = H.hh_math_fn(fn, float(x), float(y))
            assert np.float32(a).tobytes() == np.float32(b).tobytes(), (fn, x, y, a, b)
            r = ref(float(x), float(y)) if fn == 3 else ref(float(x))
            worst = max(worst, abs(a - r) / float(np.spacing(np.float32(abs(r)) if r else np.float32(1e-30))))
        assert worst <= ulps, (fn, worst)
    # special values, and erfinv (Giles) through the restated log
    for fn, x, want in [(0, -200.0, 0.0), (0, 0.0, 1.0), (0, 100.0, math.inf), (1, 1.0, 0.0), (1, 0.0, -math.inf), (1, math.inf, math.inf), (2, 10.0, 1.0), (2, -10.0, -1.0),
                        (4, 1.0, 0.0), (4, -1.0, math.pi), (6, 0.0, 0.0)]:
        a = L.orc_math_fn(fn, x, 0.0); b = H.hh_math_fn(fn, x, 0.0)
        assert a == b and abs(a - want) <= 1e-6 * max(1.0, abs(want)) if math.isfinite(want) else (a == want and b == want), (fn, x, a, b)
    assert abs(L.orc_math_fn(1, 1e-40, 0.0) - math.log(1e-40)) < 1e-4 and L.orc_math_fn(1, 1e-40, 0.0) == H.hh_math_fn(1, 1e-40, 0.0)
    for x in rng.uniform(-0.999, 0.999, 2000).astype(np.float32):
        a = L.orc_math_fn(6, float(x), 0.0); assert a == H.hh_math_fn(6, float(x), 0.0) and abs(math.erf(a) - float(x)) < 2e-6


def test_elementary_functions_against_float64_on_a_million_arguments(O):
    """The gate that keeps a coefficient typo SHARED by product and oracle from passing: har_math.h / har_bsdf.h (host build) and orc_math.h / orc_bsdf.h are each held,
    on their own, to NumPy / SciPy float64 on 10^6 arguments per function -- exp, log, erf, erfinv, atan2, acos, tan, sin, cos -- with the ulp bounds DESIGN.md
    quotes (measured maxima: exp 1.0, log 0.8, erf 1.0, acos 1.3, atan2 3.1 ulp).  Neither side is compared with the other here."""
    import scipy.special as sp
    L = O.lib(); L.orc_math_fn_array.restype = None
    L.orc_math_fn_array.argtypes = [C.c_int, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p]
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_math_fn_array.restype = None; H.hh_math_fn_array.argtypes = [C.c_int, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p]
    rng = np.random.default_rng(77)
    n = 1_000_000
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    u = rng.uniform(-1, 1, n)
    cases = {     # fn: (x, y, float64 reference, ulp bound, absolute floor below which the error is measured against that magnitude)
        0: (f32(rng.uniform(-86, 88, n)), None, np.exp, 1.5, 0.0),
        1: (f32(np.exp(rng.uniform(-87, 88, n))), None, np.log, 1.5, 0.0),
        2: (f32(rng.uniform(-4.5, 4.5, n)), None, sp.erf, 1.5, 0.0),
        3: (f32(rng.uniform(-1, 1, n)), f32(rng.uniform(-1, 1, n)), np.arctan2, 4.0, 0.0),
        4: (f32(u), None, np.arccos, 2.0, 0.0),
        5: (f32(rng.uniform(-1.5, 1.5, n)), None, np.tan, 4.0, 0.0),          # sin / cos of the shared reduction: 3.2 ulp measured
        6: (f32(np.concatenate([rng.uniform(-0.999, 0.999, n // 2), np.sign(u[:n // 2]) * (1 - np.exp(rng.uniform(-14, -1, n // 2)))])), None, sp.erfinv, 6.0, 0.0),
        7: (f32(rng.uniform(-30, 30, n)), None, np.sin, 2.0, 1e-3),
        8: (f32(rng.uniform(-30, 30, n)), None, np.cos, 2.0, 1e-3),
    }
    names = {0: "exp", 1: "log", 2: "erf", 3: "atan2", 4: "acos", 5: "tan", 6: "erfinv", 7: "sin", 8: "cos"}
    for side, fn_array in (("oracle", L.orc_math_fn_array), ("product", H.hh_math_fn_array)):
        for fn, (x, y, ref, ulps, floor) in cases.items():
            out = np.empty(n, np.float32)
            fn_array(fn, n, O.fp(x), O.fp(y) if y is not None else None, O.fp(out))
            want = ref(x.astype(np.float64), y.astype(np.float64)) if y is not None else ref(x.astype(np.float64))
            ok = np.isfinite(want)
            ulp = np.spacing(np.maximum(np.abs(want[ok]), max(floor, 1e-37)).astype(np.float32)).astype(np.float64)
            err = np.abs(out[ok].astype(np.float64) - want[ok]) / ulp
            assert np.isfinite(out[ok]).all() and err.max() <= ulps, (side, names[fn], float(err.max()), float(x[ok][err.argmax()]))


def test_scalar_gradient_sums_do_not_depend_on_the_worker_count(O):
    """render_prb_backward adds a term to the emitter / constant-albedo slots for every vertex of every path.  A float accumulator per worker loses the small
    terms once its sum has grown (6e-4 low after 3e5 paths on one thread, 3e-3 after 1e6 -- which is how the 16-core GPU boxes failed the full-film PRB test the
    256-thread runs passed); block sums in float, totals in double: the result is the same for 1 and 8 workers."""
    res, spp = 48, 96
    sd, sensor = O.cornell_box(res, res)
    osc = O.OracleScene(sd)
    g = np.full((res, res, 3), 1.0 / (res * res * 3), np.float32)
    out = {}
    for th in (1, 8):
        g_refl, _, g_emit, _ = osc.render_prb_backward_emitters(sensor, g, seed=3, spp=spp, max_depth=6, threads=th)
        out[th] = np.concatenate([np.asarray(g_refl, np.float64).reshape(-1), np.asarray(g_emit, np.float64).reshape(-1)])
    m = np.abs(out[8]) > 0
    assert m.sum() >= 6 and np.abs(out[1][m] / out[8][m] - 1).max() < 1e-5


def test_usable_core_count_follows_affinity_and_container_quota(O):
    """the product's host code (har_cpu.h) and the oracle (default_threads) each work out how many cores the process may really use -- the affinity mask
    capped by the cgroup CPU quota (16 of 256 logical CPUs on the GPU boxes) -- and must agree with each other and with a reading of the same files here"""
    import math
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_usable_cores.restype = C.c_uint
    want = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            want = min(want, max(1, math.ceil(float(q) / float(p))))
    except OSError:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                want = min(want, max(1, math.ceil(q / p)))
        except OSError:
            pass
    assert H.hh_usable_cores() == O.lib().orc_default_threads() == want >= 1


def test_builder_threads_do_not_change_the_tree():
    """har_accel_build.cpp: the collapse (dynamic program) and the node emission run on several threads for large primitive sets -- subtrees are swept / emitted
    concurrently into ranges known from a counting pass.  The arrays must be the sequential walk's byte for byte: same node order, same records, whatever the
    thread count (HAR_BUILD_EMIT_THREADS is read once per process, hence the subprocesses)."""
    import subprocess, sys
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "import mitsuba3_amd as mi; mi.set_variant('hip_ad_rgb')\n"
            "H = C.CDLL(%r); H.hh_scene_create.restype = C.c_void_p\n"
            "for d in (mi.instanced_spheres_scene(width=16, height=16, spp=1, grid=3, n_u=100, n_v=50, flatten=True), mi.instanced_spheres_scene(width=16, height=16, spp=1, grid=4, n_u=40, n_v=20), mi.cornell_box()):\n"
            "    s = mi.load_dict(d); desc = s.desc(); err = C.create_string_buffer(256); h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value\n"
            "    out = (C.c_uint64 * 3)(); H.hh_accel_hash(h, out); info = (C.c_uint64 * 4)(); H.hh_scene_info(h, info); print(list(out), list(info))\n"
            % (ROOT, os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")))
    outs = []
    for threads in ("1", "3", "8"):
        env = dict(os.environ); env["HAR_BUILD_EMIT_THREADS"] = threads
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout)
    assert outs[0].count("\n") == 3 and "[" in outs[0]
    assert outs[0] == outs[1] == outs[2]
