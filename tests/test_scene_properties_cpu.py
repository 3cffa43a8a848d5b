"""Round-5 rows a10 / a11 / a13 / f3 on the CPU:
 * every plugin constructor rejects a property it never queries (src/core/parser.cpp:1735-1760 "unreferenced property");
 * `sampling_weight` of emitters: the DiscreteDistribution over the weights (src/render/scene.cpp:120-141, 248-279, 378-388) in the oracle and in the
   product's host-compiled shading code, against each other and against the estimator's expectation;
 * bitmap `to_uv` (src/textures/bitmap.cpp:175, 565, 847): oracle == product host code == an independent NumPy lookup;
 * spatially varying area-light radiance: on rectangles (tests/test_textured_area_light_cpu.py), refused by name elsewhere; the default BSDF of an
   emitter shape is black (src/render/shape.cpp:50-57)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_cbox(mi, res=16):
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    return d


# ------------------------------------------------------------------------------------------------ unqueried properties

def _plugin_dicts(mi, tmp_path):
    """one loadable dict per plugin constructor of mitsuba3_amd.core (the `type` -> where a bogus key goes)"""
    T = mi.ScalarTransform4f
    ply = tmp_path / "tri.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                   "property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    obj = tmp_path / "tri.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    tex = np.full((4, 4, 3), 0.5, np.float32)
    env = mi.Bitmap(np.full((4, 8, 3), 1.0, np.float32))
    return {
        "rectangle": {"type": "rectangle"}, "cube": {"type": "cube"},
        "mesh": {"type": "mesh", "positions": np.eye(3, dtype=np.float32), "faces": np.array([[0, 1, 2]], np.uint32)},
        "ply": {"type": "ply", "filename": str(ply)}, "obj": {"type": "obj", "filename": str(obj)},
        "shapegroup": {"type": "shapegroup", "a": {"type": "cube"}},
        "diffuse": {"type": "diffuse"}, "dielectric": {"type": "dielectric"}, "conductor": {"type": "conductor"}, "plastic": {"type": "plastic"},
        "roughconductor": {"type": "roughconductor"}, "roughplastic": {"type": "roughplastic"},
        "twosided": {"type": "twosided", "nested": {"type": "diffuse"}},
        "constant": {"type": "constant"}, "envmap": {"type": "envmap", "bitmap": env},
        "point": {"type": "point"}, "spot": {"type": "spot"}, "directional": {"type": "directional"},
        "hdrfilm": {"type": "hdrfilm"}, "independent": {"type": "independent"},
        "perspective": {"type": "perspective", "to_world": T().look_at([0, 0, 3], [0, 0, 0], [0, 1, 0])},
        "orthographic": {"type": "orthographic"},
        "path": {"type": "path"}, "prb": {"type": "prb"},
        "scene": {"type": "scene"},
        # nested plugins: the bogus key goes into the inner dict
        "area": ("emitter", {"type": "rectangle", "emitter": {"type": "area"}}),
        "bitmap": ("reflectance", {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex}}),
    }


def test_every_plugin_rejects_an_unqueried_property(mi, tmp_path):
    """src/core/parser.cpp:1735-1760: a property no constructor queried is an error, for EVERY plugin -- a silently ignored property is a silently
    different picture (round 4's defect: rectangle / diffuse / area / constant accepted anything)"""
    plugins = _plugin_dicts(mi, tmp_path)
    for name, spec in plugins.items():
        inner_key, d = spec if isinstance(spec, tuple) else (None, spec)
        mi.load_dict(d)                                                        # the clean dict loads
        bad = dict(d)
        if inner_key is None:
            bad["bogus_property"] = 3.0
        else:
            bad[inner_key] = dict(d[inner_key]); bad[inner_key]["bogus_property"] = 3.0
        with pytest.raises(RuntimeError, match="(?i)unreferenced"):
            mi.load_dict(bad)
        # ... and a bogus child OBJECT is only accepted where the reference's constructor walks props.objects() (shapes, scene, twosided, sensors)
        if name in ("diffuse", "dielectric", "conductor", "plastic", "roughconductor", "roughplastic", "constant", "point", "spot", "directional", "envmap"):
            bad = dict(d); bad["bogus_child"] = {"type": "rgb", "value": [0.1, 0.2, 0.3]}
            with pytest.raises(RuntimeError, match="(?i)unreferenced"):
                mi.load_dict(bad)
    # instance: needs its shapegroup, checked inside a scene
    sc = {"type": "scene", "g": {"type": "shapegroup", "a": {"type": "cube"}}, "i": {"type": "instance", "g": {"type": "ref", "id": "g"}}}
    mi.load_dict(sc)
    sc["i"] = dict(sc["i"]); sc["i"]["bogus_property"] = 1
    with pytest.raises(RuntimeError, match="(?i)unreferenced"):
        mi.load_dict(sc)


def test_known_but_unimplemented_values_are_refused_not_ignored(mi):
    for d in ({"type": "cube", "flip_normals": True}, {"type": "rectangle", "face_normals": True},
              {"type": "twosided", "nested": {"type": "diffuse"}, "allow_transmission": True},
              {"type": "diffuse", "reflectance": {"type": "bitmap", "data": np.zeros((2, 2, 3), np.float32), "format": "fp16"}}):
        with pytest.raises(RuntimeError, match="not implemented"):
            mi.load_dict(d)
    # the neutral values are accepted
    mi.load_dict({"type": "cube", "flip_normals": False, "silhouette_sampling_weight": 2.0})
    mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "data": np.zeros((2, 2, 3), np.float32), "raw": True, "accel": False, "format": "auto"}})


def test_area_light_placement_and_textured_radiance(mi):
    with pytest.raises(RuntimeError, match="to_world"):            # area.cpp:66-69
        mi.load_dict({"type": "rectangle", "emitter": {"type": "area", "to_world": mi.ScalarTransform4f()}})
    # area.cpp:74,133-165: a bitmap radiance is importance-sampled and mapped through Shape::eval_parameterization -- built for rectangles
    # (tests/test_textured_area_light_cpu.py), refused by name on triangle meshes and for the other emitters' emissive parameters
    mi.load_dict({"type": "rectangle", "emitter": {"type": "area", "radiance": {"type": "bitmap", "data": np.ones((4, 4, 3), np.float32)}}})
    with pytest.raises(RuntimeError, match="eval_parameterization"):
        mi.load_dict({"type": "scene", "c": {"type": "cube", "emitter": {"type": "area", "radiance": {"type": "bitmap", "data": np.ones((4, 4, 3), np.float32)}}}})
    with pytest.raises(RuntimeError, match="spatially varying"):
        mi.load_dict({"type": "constant", "radiance": {"type": "bitmap", "data": np.ones((4, 4, 3), np.float32)}})
    with pytest.raises(RuntimeError, match="single Emitter"):      # shape.cpp:25-27
        mi.load_dict({"type": "rectangle", "e1": {"type": "area"}, "e2": {"type": "area"}})


def test_default_bsdf_of_an_emitter_shape_is_black(mi):
    """Shape(props), src/render/shape.cpp:50-57: `props2.set("reflectance", 0.f)` when the shape carries an emitter"""
    sc = mi.load_dict({"type": "scene", "lamp": {"type": "rectangle", "emitter": {"type": "area"}}, "wall": {"type": "rectangle"}})
    by_key = {m["key"]: sc.bsdfs[m["bsdf"]] for m in sc.meshes}
    assert np.all(by_key["lamp"].value == 0.0) and np.allclose(by_key["wall"].value, 0.5)


# ------------------------------------------------------------------------------------------------ sampling_weight

def two_light_scene(mi, w0=1.0, w1=3.0, res=16, sky=None):
    T = mi.ScalarTransform4f
    d = _small_cbox(mi, res)
    d["light"]["emitter"]["sampling_weight"] = w0
    d["lamp2"] = {"type": "rectangle", "to_world": T().translate([-0.5, 0.2, -0.3]).rotate([0, 1, 0], 70).scale([0.15, 0.25, 0.2]), "bsdf": {"type": "ref", "id": "white"},
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 6.0, 9.0]}, "sampling_weight": w1}}
    if sky is not None:
        d.pop("ceiling")
        d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.3, 0.4, 0.5]}, "sampling_weight": sky}
    return d


def test_sampling_weight_validation(mi):
    with pytest.raises(RuntimeError, match="non-negative"):
        mi.load_dict(two_light_scene(mi, 1.0, -1.0))
    sc = mi.load_dict(two_light_scene(mi, 0.0, 0.0))
    with pytest.raises(Exception, match="no probability mass"):          # distr_1d.h:259-260, raised by the lowering
        from tests.test_emitters_cpu import _harness_render
        from oracle import oracle as O
        _harness_render(mi, O, sc, sc.sensors()[0].har, 0, 0, 1, 2)


def test_emitter_distribution_tables(mi, O):
    """DiscreteDistribution(values, n) -> compute_cdf_scalar (distr_1d.h:236-266): double running sums rounded per entry, m_valid = first / last bin with mass;
    the oracle's selection visits emitters with frequency w_i / sum and re-uses the sample uniformly"""
    L = O.lib()
    w = np.array([0.0, 1.0, 3.0, 0.0, 0.5, 0.0], np.float32); n = 200000
    u = ((np.arange(n) + 0.5) / n).astype(np.float32)
    idx = np.zeros(n, np.uint32); reused = np.zeros(n, np.float32); pmf = np.zeros(n, np.float32)
    L.orc_discrete_sample_reuse.argtypes = [O.c_f32p, C.c_uint32, C.c_uint32, O.c_f32p, O.c_u32p, O.c_f32p, O.c_f32p]; L.orc_discrete_sample_reuse.restype = None
    L.orc_discrete_sample_reuse(O.fp(w), len(w), n, O.fp(u), idx.ctypes.data_as(O.c_u32p), O.fp(reused), O.fp(pmf))
    freq = np.bincount(idx, minlength=len(w)) / n
    assert np.allclose(freq, w / w.sum(), atol=2e-5) and freq[0] == 0 and freq[3] == 0 and freq[5] == 0
    assert np.allclose(pmf, (w / w.sum())[idx], rtol=1e-6)
    assert reused.min() >= 0.0 and reused.max() <= 1.0 + 1e-6
    for k in (1, 2, 4):                                                # the re-used sample is uniform again inside every bin
        r = reused[idx == k]; assert abs(r.mean() - 0.5) < 5e-3 and r.max() - r.min() > 0.99


@pytest.mark.parametrize("weights", [(1.0, 3.0, None), (0.25, 1.0, None), (2.0, 2.0, None), (1.0, 3.0, 0.5), (1.0, 0.0, None)])
def test_sampling_weight_product_host_code_matches_oracle(mi, O, weights):
    """the product's shading headers (har_path.h, compiled for the host) against the oracle on scenes with non-uniform emitter selection: `path` and the
    primal pass of `prb`, bit-close films.  (2, 2): weights that are equal but not 1 still build the distribution (scene.cpp:123-128);
    (1, 0): an emitter that is never chosen but still hit by BSDF samples (its MIS pmf is 0)."""
    from tests.test_emitters_cpu import _harness_render
    scene = mi.load_dict(two_light_scene(mi, weights[0], weights[1], sky=weights[2]))
    assert [e["sampling_weight"] for e in scene.emitters][:2] == [weights[0], weights[1]]
    osc, sensor = O.scene_from_product(scene)
    for mode, md in ((0, 6), (1, 5)):
        film = _harness_render(mi, O, scene, sensor, mode, 3, 16, md)
        ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=16, max_depth=md, raw=True, threads=2)
        a, b = O.develop(film), O.develop(ref)
        assert np.isfinite(a).all() and np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5


def test_sampling_weight_changes_the_stream_not_the_expectation(mi, O):
    """weights 1 : 3 draw a DIFFERENT sample stream than uniform selection (so ignoring the property silently is a different picture), and the same
    expected image (the estimator divides by the selection probability): per-pixel difference of two independent estimates inside their noise"""
    a_sc, sensor = O.scene_from_product(mi.load_dict(two_light_scene(mi, 1.0, 3.0, res=12)))
    b_sc, _ = O.scene_from_product(mi.load_dict(two_light_scene(mi, 1.0, 1.0, res=12)))
    a, _ = a_sc.render_path(sensor, seed=5, spp=1024, max_depth=4, threads=4)
    b, _ = b_sc.render_path(sensor, seed=5, spp=1024, max_depth=4, threads=4)
    assert np.abs(a - b).max() > 1e-4                              # not the same samples
    assert abs(a.mean() / b.mean() - 1.0) < 0.01                   # the same expectation
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 0.08


def test_single_weighted_emitter_is_the_uniform_picture(mi, O):
    """one emitter with sampling_weight 3: JIT variants still multiply by pdf_emitter = w * (1 / w) (scene.cpp:326-338), which is 1 for w = 3"""
    d = _small_cbox(mi, 12); d["light"]["emitter"]["sampling_weight"] = 3.0
    a_sc, sensor = O.scene_from_product(mi.load_dict(d))
    b_sc, _ = O.scene_from_product(mi.load_dict(_small_cbox(mi, 12)))
    a, _ = a_sc.render_path(sensor, seed=2, spp=8, max_depth=5, raw=True); b, _ = b_sc.render_path(sensor, seed=2, spp=8, max_depth=5, raw=True)
    assert np.array_equal(a, b)


def test_scalar_driver_with_weighted_emitters_matches_oracle(mi, O):
    """config 1's scalar driver with a distribution: the non-JIT predicate of DiscreteDistribution::sample (distr_1d.h:126-127)"""
    scene = mi.load_dict(two_light_scene(mi, 1.0, 3.0, res=12))
    osc, sensor = O.scene_from_product(scene)
    mi.set_variant("scalar_rgb")
    try:
        from mitsuba3_amd import core
        img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 4, 8, threads=1)      # one worker: the oracle's block size
    finally:
        mi.set_variant("hip_ad_rgb")
    ref = osc.render_path_scalar(sensor, seed=4, spp=8, max_depth=8)[0]
    assert np.isfinite(img).all() and img.mean() > 0
    assert np.linalg.norm(img - ref) / np.linalg.norm(ref) < 1e-5


# ------------------------------------------------------------------------------------------------ bitmap to_uv

def textured_scene(mi, to_uv, res=16, filter_type="bilinear", wrap_mode="repeat", tex_res=8):
    rng = np.random.default_rng(11)
    tex = rng.uniform(0.05, 0.95, (tex_res, tex_res + 3, 3)).astype(np.float32)
    d = _small_cbox(mi, res)
    refl = {"type": "bitmap", "data": tex, "filter_type": filter_type, "wrap_mode": wrap_mode}
    if to_uv is not None:
        refl["to_uv"] = to_uv
    d["white"] = {"type": "diffuse", "reflectance": refl}
    return d, tex


def test_to_uv_argument_checks(mi):
    d, _ = textured_scene(mi, "not a transform")
    with pytest.raises(RuntimeError, match="ScalarTransform3f"):
        mi.load_dict(d)
    d, _ = textured_scene(mi, mi.ScalarTransform3f().scale([0.0, 1.0]))
    with pytest.raises(RuntimeError, match="singular"):
        mi.load_dict(d)


def test_transform3f_chain_order(mi):
    """T().translate(t).rotate(a).scale(s) = translate * rotate * scale: the scale acts on a point first (transform.h operator*)"""
    T = mi.ScalarTransform3f
    m = T().translate([0.25, -0.5]).rotate(90).scale([2.0, 3.0]).matrix
    p = m @ np.array([1.0, 1.0, 1.0])
    assert np.allclose(p[:2], [0.25 - 3.0, -0.5 + 2.0], atol=1e-6)
    inv = T(m).inverse().matrix
    assert np.allclose(inv @ m, np.eye(3), atol=1e-5)


@pytest.mark.parametrize("filter_type,wrap_mode", [("bilinear", "repeat"), ("nearest", "mirror"), ("bilinear", "clamp")])
def test_to_uv_lookup_against_numpy(mi, O, filter_type, wrap_mode):
    """BitmapTexture::eval with `to_uv` against an independent float64 NumPy lookup at (to_uv * uv), through the product's host BSDF code and the oracle
    (a diffuse BSDF's value at normal incidence is rho / pi * cos)"""
    T = mi.ScalarTransform3f
    tuv = T().translate([0.37, -0.21]).rotate(33.0).scale([2.5, 0.75])
    d, tex = textured_scene(mi, tuv, filter_type=filter_type, wrap_mode=wrap_mode)
    scene = mi.load_dict(d)
    assert scene.texture_to_uv[scene.bsdfs[0].tex_index] is not None or any(t is not None for t in scene.texture_to_uv)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    L.hh_bsdf_eval_pdf.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p, O.c_f32p, O.c_f32p]
    bi = [b.index for b in scene.bsdfs if b.texture is not None][0]
    H, W, _ = tex.shape
    M = np.asarray(tuv.matrix, np.float64)
    rng = np.random.default_rng(3)

    def wrap(i, n):
        if wrap_mode == "clamp":
            return min(max(i, 0), n - 1)
        r, m = divmod(i, n)
        return n - 1 - m if (wrap_mode == "mirror" and r % 2) else m

    worst = 0.0
    for uv in rng.uniform(-1.5, 2.5, (200, 2)):
        val = np.zeros(3, np.float32); pdf = np.zeros(1, np.float32)
        L.hh_bsdf_eval_pdf(h, bi, O.fp(O.f32([0, 0, 1])), O.fp(O.f32(uv)), O.fp(O.f32([0, 0, 1])), O.fp(val), O.fp(pdf))
        q = M @ np.array([np.float32(uv[0]), np.float32(uv[1]), 1.0])
        if filter_type == "nearest":
            x, y = int(np.floor(q[0] * W)), int(np.floor(q[1] * H))
            if abs(q[0] * W - round(q[0] * W)) < 1e-3 or abs(q[1] * H - round(q[1] * H)) < 1e-3:
                continue                                           # float32 vs float64 may land on different sides of a texel edge
            ref = tex[wrap(y, H), wrap(x, W)].astype(np.float64)
        else:
            px, py = q[0] * W - 0.5, q[1] * H - 0.5
            x0, y0 = int(np.floor(px)), int(np.floor(py)); fx, fy = px - x0, py - y0
            t = lambda yy, xx: tex[wrap(yy, H), wrap(xx, W)].astype(np.float64)
            ref = (1 - fy) * ((1 - fx) * t(y0, x0) + fx * t(y0, x0 + 1)) + fy * ((1 - fx) * t(y0 + 1, x0) + fx * t(y0 + 1, x0 + 1))
        worst = max(worst, float(np.abs(val * np.pi - ref).max()))
    assert worst < 2e-4, worst
    L.hh_scene_destroy.argtypes = [C.c_void_p]; L.hh_scene_destroy(h)


@pytest.mark.parametrize("mode", [0, 1])
def test_to_uv_product_host_code_matches_oracle(mi, O, mode):
    from tests.test_emitters_cpu import _harness_render
    T = mi.ScalarTransform3f
    d, _ = textured_scene(mi, T().translate([0.1, 0.3]).rotate(-20.0).scale([3.0, 2.0]))
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    md = 6 if mode == 0 else 5
    film = _harness_render(mi, O, scene, sensor, mode, 9, 16, md)
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=9, spp=16, max_depth=md, raw=True, threads=2)
    a, b = O.develop(film), O.develop(ref)
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5
    # ... and the transform matters: the untransformed texture is a different picture
    d0, _ = textured_scene(mi, None)
    osc0, _ = O.scene_from_product(mi.load_dict(d0))
    ref0, _ = (osc0.render_path if mode == 0 else osc0.render_prb)(sensor, seed=9, spp=16, max_depth=md, raw=True, threads=2)
    assert np.linalg.norm(O.develop(ref0) - b) / np.linalg.norm(b) > 1e-2


def test_to_uv_texel_gradients_oracle_vs_finite_differences(mi, O):
    """PRB gradient w.r.t. the texels of a transformed bitmap (the adjoint scatters through the same taps): oracle vs a central finite difference of the
    oracle's own primal render along a random texel direction"""
    T = mi.ScalarTransform3f
    d, tex = textured_scene(mi, T().rotate(25.0).scale([1.7, 1.3]), res=8, tex_res=4)
    d["integrator"] = {"type": "prb", "max_depth": 4}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    ti = [b.tex_index for b in scene.bsdfs if b.texture is not None][0]
    rng = np.random.default_rng(5); direction = rng.standard_normal(tex.shape).astype(np.float32)
    adj = rng.uniform(0.5, 1.5, (8, 8, 3)).astype(np.float32)
    out = osc.render_prb_backward(sensor, adj, seed=1, spp=256, max_depth=4, threads=4)
    g_tex = out[1][ti]
    analytic = float((g_tex * direction).sum())
    eps = 2e-2
    imgs = []
    for sgn in (+1, -1):
        osc.set_texture(ti, (tex + sgn * eps * direction).astype(np.float32))
        img, _ = osc.render_prb(sensor, seed=1, spp=256, max_depth=4, threads=4)
        imgs.append(img)
    osc.set_texture(ti, tex)
    fd = float(((imgs[0] - imgs[1]) / (2 * eps) * adj).sum())
    assert abs(analytic - fd) / abs(fd) < 0.05, (analytic, fd)


# ------------------------------------------------------------------------------------------------ the reference's own scene tests (src/render/tests/test_scene.py)

def _three_emitter_scene(mi, weights):
    """test_scene.py:162-172 with a rectangle for the sphere (triangle scenes only): an area light, a point light and a constant environment"""
    return mi.load_dict({'type': 'scene',
                         'shape': {'type': 'rectangle', 'emitter': {'type': 'area', 'sampling_weight': weights[0]}},
                         'emitter_0': {'type': 'point', 'sampling_weight': weights[1]},
                         'emitter_1': {'type': 'constant', 'sampling_weight': weights[2]}})


def _hh(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_scene_sample_emitter.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, C.c_int, O.c_u32p, O.c_f32p, O.c_f32p]
    L.hh_scene_pdf_emitter.argtypes = [C.c_void_p, C.c_uint32, O.c_u32p, O.c_f32p]
    L.hh_scene_set_emitter_weights.argtypes = [C.c_void_p, O.c_f32p, C.c_uint32, C.c_char_p, C.c_int]
    return L


def _hh_sample(L, O, h, sample):
    s = O.f32(sample).reshape(-1); n = s.size
    idx = np.zeros(n, np.uint32); w = np.zeros(n, np.float32); r = np.zeros(n, np.float32)
    L.hh_scene_sample_emitter(h, n, O.fp(s), 1, idx.ctypes.data_as(O.c_u32p), O.fp(w), O.fp(r))
    return idx, w, r


@pytest.mark.parametrize("weights", [[1.0, 1.0, 1.0], [1.3, 3.8, 0.0]])
def test_reference_emitter_pdf_and_sampling(mi, O, weights):
    """test_scene.py:162-201 (test05_emitter_pdf, test06_emitter_sampling): pdf_emitter(i) = w_i / sum; sample_emitter(0.75) agrees with a DiscreteDistribution over the
    weights (hand-evaluated here: index, 1 / pmf, re-used sample) -- on the oracle and on the product's host code"""
    scene = _three_emitter_scene(mi, weights)
    osc, _ = O.scene_from_product(scene)
    L = _hh(O); desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    pdf = np.array(weights) / np.sum(weights)
    got_p = np.zeros(3, np.float32); L.hh_scene_pdf_emitter(h, 3, np.arange(3, dtype=np.uint32).ctypes.data_as(O.c_u32p), O.fp(got_p))
    assert np.allclose(osc.pdf_emitter([0, 1, 2]), pdf, rtol=1e-6) and np.allclose(got_p, pdf, rtol=1e-6)
    sample = 0.75
    cdf = np.cumsum(weights) / np.sum(weights)
    ref_index = int(np.searchsorted(cdf, sample, side="left")) if weights != [1.0, 1.0, 1.0] else min(int(sample * 3), 2)
    ref_pmf = pdf[ref_index]; ref_reused = (sample - (cdf[ref_index - 1] if ref_index else 0.0)) / ref_pmf
    for index, weight, reused in (osc.sample_emitter([sample]), _hh_sample(L, O, h, [sample])):
        assert int(index[0]) == ref_index and np.isclose(weight[0], 1.0 / ref_pmf, rtol=1e-6) and np.isclose(reused[0], ref_reused, rtol=1e-5, atol=1e-6)
    # ... and on a sweep of samples the two implementations agree exactly
    u = ((np.arange(4097) + 0.5) / 4097).astype(np.float32)
    a = osc.sample_emitter(u); b = _hh_sample(L, O, h, u)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_reference_emitter_weight_update(mi, O):
    """test_scene.py:204-233 (test07_emitter_weight_update): editing sampling_weight through traverse rebuilds the distribution"""
    scene = mi.load_dict({'type': 'scene', 'emitter_0': {'type': 'point', 'sampling_weight': 2.0}, 'emitter_1': {'type': 'constant', 'sampling_weight': 1.0},
                          'emitter_2': {'type': 'directional', 'sampling_weight': 0.5}})
    params = mi.traverse(scene)
    import torch
    for k, v in (('emitter_0.sampling_weight', 0.8), ('emitter_1.sampling_weight', 0.05), ('emitter_2.sampling_weight', 1.2)):
        assert k in params
        params[k] = torch.tensor([v])
    params.update()
    weights = [e["sampling_weight"] for e in scene.emitters]
    assert np.allclose(weights, [0.8, 0.05, 1.2])
    osc, _ = O.scene_from_product(scene)                      # a freshly lowered scene with the new weights
    L = _hh(O); desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    pdf = np.array(weights) / np.sum(weights)
    assert np.allclose(osc.pdf_emitter([0, 1, 2]), pdf, rtol=1e-6)
    cdf = np.cumsum(weights) / np.sum(weights); ref_index = int(np.searchsorted(cdf, 0.75)); ref_pmf = pdf[ref_index]
    for index, weight, reused in (osc.sample_emitter([0.75]), _hh_sample(L, O, h, [0.75])):
        assert int(index[0]) == ref_index and np.isclose(weight[0], 1 / ref_pmf, rtol=1e-6) and np.isclose(reused[0], (0.75 - cdf[ref_index - 1]) / ref_pmf, rtol=1e-5)
    # the in-place updates (oracle: orc_scene_set_emitter_weights; product host code: the path of har_scene_set_emitter_sampling_weights) equal the fresh lowering
    old = mi.load_dict({'type': 'scene', 'emitter_0': {'type': 'point', 'sampling_weight': 2.0}, 'emitter_1': {'type': 'constant', 'sampling_weight': 1.0},
                        'emitter_2': {'type': 'directional', 'sampling_weight': 0.5}})
    osc2, _ = O.scene_from_product(old); osc2.set_emitter_weights(weights)
    d2 = old.desc(); h2 = C.c_void_p(L.hh_scene_create(C.byref(d2), err, 256)); assert h2
    assert L.hh_scene_set_emitter_weights(h2, O.fp(O.f32(weights)), 3, err, 256) == 0
    u = ((np.arange(1025) + 0.5) / 1025).astype(np.float32)
    ref = osc.sample_emitter(u)
    for got in (osc2.sample_emitter(u), _hh_sample(L, O, h2, u)):
        assert all(np.array_equal(x, y) for x, y in zip(got, ref))
    assert L.hh_scene_set_emitter_weights(h2, O.fp(O.f32([0, 0, 0])), 3, err, 256) == 1 and b"no probability mass" in err.value


def test_reference_to_uv_is_a_traversable_parameter(mi, O):
    """src/textures/tests/test_bitmap.py:270-281 (test08_to_uv): `to_uv` shows up in traverse with the value it was given; an update re-lowers the lookup"""
    T = mi.ScalarTransform3f
    transform = T().translate([2, 4]).scale([3, 9]).rotate(45)
    d, tex = textured_scene(mi, transform)
    scene = mi.load_dict(d); params = mi.traverse(scene)
    key = "white.reflectance.to_uv"
    assert key in params and np.allclose(params[key].cpu().numpy(), transform.matrix, atol=1e-6)
    import torch
    new = T().rotate(-10).scale([1.5, 0.5])
    params[key] = torch.tensor(new.matrix); params.update()
    fresh_d, _ = textured_scene(mi, new); fresh = mi.load_dict(fresh_d)
    assert np.allclose(scene.texture_to_uv[scene.bsdfs[0].tex_index if scene.bsdfs[0].texture is not None else [b for b in scene.bsdfs if b.texture is not None][0].tex_index],
                       fresh.texture_to_uv[[b for b in fresh.bsdfs if b.texture is not None][0].tex_index])
    a, sensor = O.scene_from_product(scene); b, _ = O.scene_from_product(fresh)
    ia, _ = a.render_path(sensor, seed=2, spp=4, max_depth=4, raw=True); ib, _ = b.render_path(sensor, seed=2, spp=4, max_depth=4, raw=True)
    assert np.array_equal(ia, ib)


def test_traverse_names_of_twosided_children(mi):
    """mi.traverse() names the BSDFs nested in a `twosided` the way TwoSidedBRDF::traverse registers them (twosided.cpp:106-109): '<twosided>.brdf_0.*' and -- when the back
    side is a BSDF of its own -- '<twosided>.brdf_1.*'; a twosided with ONE nested BSDF has no brdf_1 entries (util.py:320-334 walks an object once).  Two twosided BSDFs
    whose children carry the same dict key ('back') must not share a parameter name: round 6's randomised suite found 'back.specular_reflectance.value' naming the
    roughconductor under one floor AND the specular colour of a roughplastic under another shape"""
    T = mi.ScalarTransform4f
    rgb = lambda *v: {"type": "rgb", "value": list(v)}
    d = {"type": "scene", "integrator": {"type": "prb", "max_depth": 3},
         "sensor": {"type": "perspective", "fov": 45, "to_world": T().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": 8, "height": 8, "rfilter": {"type": "box"}, "pixel_format": "rgb"}, "sampler": {"type": "independent", "sample_count": 4}},
         "wall": {"type": "twosided", "m": {"type": "diffuse", "reflectance": rgb(0.1, 0.2, 0.3)}},
         "floor": {"type": "rectangle", "bsdf": {"type": "twosided", "front": {"type": "diffuse", "reflectance": rgb(0.5, 0.5, 0.5)},
                                                  "back": {"type": "roughconductor", "alpha": 0.2, "specular_reflectance": rgb(0.9, 0.8, 0.7)}}},
         "sheet": {"type": "rectangle", "to_world": T().translate([0, 0, 1]),
                   "bsdf": {"type": "twosided", "front": {"type": "roughplastic", "alpha": 0.3, "diffuse_reflectance": rgb(0.2, 0.2, 0.2)},
                            "back": {"type": "roughplastic", "alpha": 0.1, "diffuse_reflectance": rgb(0.4, 0.4, 0.4), "specular_reflectance": rgb(0.6, 0.6, 0.6)}}},
         "side": {"type": "rectangle", "to_world": T().translate([2, 0, 0]), "bsdf": {"type": "ref", "id": "wall"}},
         "light": {"type": "point", "position": [0, 0, 3], "intensity": rgb(1, 1, 1)}}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    keys = set(params.keys())
    for k in ("wall.brdf_0.reflectance.value", "floor.bsdf.brdf_0.reflectance.value", "floor.bsdf.brdf_1.specular_reflectance.value", "floor.bsdf.brdf_1.alpha.value",
              "floor.bsdf.brdf_1.eta.value", "floor.bsdf.brdf_1.k.value", "sheet.bsdf.brdf_0.diffuse_reflectance.value", "sheet.bsdf.brdf_0.alpha",
              "sheet.bsdf.brdf_1.diffuse_reflectance.value", "sheet.bsdf.brdf_1.specular_reflectance.value", "sheet.bsdf.brdf_1.alpha"):
        assert k in keys, (k, sorted(keys))
    assert not any(k.startswith(("back.", "front.", "m.")) or ".brdf_1." in k and k.startswith("wall") for k in keys), sorted(keys)
    # the two specular colours are two parameters with their own values
    assert np.allclose(np.asarray(params["floor.bsdf.brdf_1.specular_reflectance.value"].cpu()), [0.9, 0.8, 0.7])
    assert np.allclose(np.asarray(params["sheet.bsdf.brdf_1.specular_reflectance.value"].cpu()), [0.6, 0.6, 0.6])
    # the colour table and the table of the other BSDF parameters never name the same key
    assert not (set(scene._param_keys()) & set(scene._bsdf_param_keys()))


def test_traverse_mesh_entries_and_unknown_keys(mi):
    """Mesh::traverse (src/render/mesh.cpp:822-843) names the vertex positions '<shape>.positions' (an N x 3 tensor) and also registers 'faces' / 'texcoords': the
    positions are the updatable / differentiable entry here, the other two are shown and refused on write like a read-only parameter (util.py:59-60); a key that
    traverse() did not register is a KeyError on write (util.py:57), not a silently ignored entry"""
    import torch
    T = mi.ScalarTransform4f
    P = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32); F = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    d = {"type": "scene", "integrator": {"type": "prb", "max_depth": 3},
         "sensor": {"type": "perspective", "fov": 45, "to_world": T().look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": 8, "height": 8, "rfilter": {"type": "box"}, "pixel_format": "rgb"}, "sampler": {"type": "independent", "sample_count": 4}},
         "quad": {"type": "mesh", "positions": P, "faces": F, "texcoords": uv, "bsdf": {"type": "diffuse"}},
         "bare": {"type": "mesh", "positions": P + np.float32(3.0), "faces": F, "bsdf": {"type": "diffuse"}},
         "light": {"type": "point", "position": [0, 0, 3], "intensity": {"type": "rgb", "value": [1, 1, 1]}}}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    assert tuple(params["quad.positions"].shape) == (4, 3) and np.array_equal(params["quad.positions"].cpu().numpy(), P)
    assert np.array_equal(params["quad.faces"].cpu().numpy(), F) and np.array_equal(params["quad.texcoords"].cpu().numpy(), uv)
    assert "bare.faces" in params and "bare.texcoords" not in params and not any(k.endswith("vertex_positions") for k in params)
    with pytest.raises(Exception, match="read-only"):
        params["quad.faces"] = params["quad.faces"].clone()
    with pytest.raises(KeyError):
        params["quad.vertex_positions"] = torch.zeros(12)
    params.update({"nonsense": torch.zeros(3)})                  # update(values) skips names that are not parameters (util.py:210-213)
    assert "nonsense" not in params
    # positions: N x 3 or flat on write
    params["quad.positions"] = torch.as_tensor((P * np.float32(0.5)).reshape(-1)); params.update()
    assert np.array_equal(scene.meshes[scene._position_keys()["quad.positions"]]["V"][:, :3], P * np.float32(0.5))


def test_traverse_shows_the_sensors_plain_parameters(mi):
    """what ProjectiveCamera / Sensor / Film register next to `to_world` (sensor.h:135-141,206-210, film.cpp:55-57) is readable under the reference's names and read-only"""
    d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = 48; f["height"] = 32; f["crop_offset_x"] = 4; f["crop_offset_y"] = 2; f["crop_width"] = 20; f["crop_height"] = 10
    params = mi.traverse(mi.load_dict(d))
    assert params["sensor.film.size"].tolist() == [48, 32] and params["sensor.film.crop_size"].tolist() == [20, 10] and params["sensor.film.crop_offset"].tolist() == [4, 2]
    assert abs(float(params["sensor.near_clip"]) - 0.001) < 1e-9 and float(params["sensor.far_clip"]) == 100.0
    with pytest.raises(Exception, match="read-only"):
        params["sensor.film.size"] = params["sensor.film.size"]
    params["sensor.to_world"] = params["sensor.to_world"].clone(); params.update()          # the placement stays updatable


def test_rectangle_to_world_is_an_updatable_parameter(mi, O):
    """Rectangle::traverse registers 'to_world' (src/shapes/rectangle.cpp:197-200; the reference's tests write params['shape.to_world'], test_rectangle.py:243-406): writing it +
    params.update() re-bakes the vertex records, the winding (a mirroring transform) and, for a rectangle that carries an area light, its sampling record -- the scene then equals
    a freshly loaded one with that transform, bit for bit (oracle render of the host mirrors)"""
    import torch
    T = mi.ScalarTransform4f
    d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = 24; f["height"] = 24
    moves = {"floor": T().translate([0.0, -0.8, 0.1]).rotate([1, 0, 0], -80.0).scale([1.2, 0.9, 1.0]),
             "light": T().translate([0.1, 0.95, 0.05]).rotate([1, 0, 0], 90.0).scale([0.3, -0.2, 1.0]),          # (a mirrored light: the winding flips)
             "back": T().translate([0.0, 0.0, -1.0]).scale([1.0, 1.1, 1.0])}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    for k in moves:
        assert tuple(params[k + ".to_world"].shape) == (4, 4)
    fresh = dict(d)
    for k, t in moves.items():
        params[k + ".to_world"] = torch.as_tensor(np.asarray(t.matrix, np.float32))
        fresh[k] = dict(d[k]); fresh[k]["to_world"] = T(np.concatenate([np.asarray(t.matrix, np.float64).ravel(), np.linalg.inv(np.asarray(t.matrix, np.float64)).T.ravel()]).astype(np.float32))
    params.update()
    want = mi.load_dict(fresh)
    for a, b in zip(scene.meshes, want.meshes):
        assert np.array_equal(a["V"], b["V"]) and np.array_equal(a["F"], b["F"]), a["key"]
    for a, b in zip(scene.emitters, want.emitters):
        for key in ("to_world", "normal", "inv_area"):
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key
    o1, s1 = O.scene_from_product(scene); o2, s2 = O.scene_from_product(want)
    a, st1 = o1.render_path(s1, seed=3, spp=4, max_depth=5, threads=1); b, st2 = o2.render_path(s2, seed=3, spp=4, max_depth=5, threads=1)
    assert np.array_equal(a, b) and st1.vertices == st2.vertices
    assert np.allclose(mi.traverse(scene)["light.to_world"].cpu().numpy(), np.asarray(moves["light"].matrix, np.float32))
    with pytest.raises(RuntimeError, match="singular"):
        params["floor.to_world"] = torch.zeros((4, 4)); params.update()


def test_x_fov_is_a_parameter_of_the_perspective_sensor(mi, O):
    """PerspectiveCamera::traverse registers 'x_fov' (perspective.cpp:157): the value is parse_fov's (sensor.cpp:142-195: `fov_axis` x / y / diagonal / smaller / larger and
    `focal_length` at the film's aspect ratio); writing it re-lowers the projection -- the sensor then equals one created with that horizontal angle"""
    import math, torch
    def cbox(w, h, **sensor):
        d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = w; f["height"] = h
        for k in ("fov", "fov_axis"):
            d["sensor"].pop(k, None)
        d["sensor"].update(sensor)
        return d
    for w, h, kw, want in ((40, 20, dict(fov=40.0), 40.0),
                           (40, 20, dict(fov=40.0, fov_axis="y"), math.degrees(2 * math.atan(math.tan(math.radians(20.0)) * 2.0))),
                           (40, 20, dict(fov=40.0, fov_axis="smaller"), math.degrees(2 * math.atan(math.tan(math.radians(20.0)) * 2.0))),
                           (20, 40, dict(fov=40.0, fov_axis="larger"), math.degrees(2 * math.atan(math.tan(math.radians(20.0)) * 0.5))),
                           (30, 30, dict(fov=50.0, fov_axis="diagonal"), math.degrees(2 * math.atan(math.tan(math.radians(25.0)) / math.sqrt(2.0)))),
                           (36, 24, dict(focal_length="50mm"), math.degrees(2 * math.atan(18.0 / 50.0)))):
        scene = mi.load_dict(cbox(w, h, **kw))
        got = float(mi.traverse(scene)["sensor.x_fov"])
        assert abs(got - want) < 1e-4, (kw, got, want)
        ref = mi.load_dict(cbox(w, h, fov=want)).sensors()[0]
        assert np.allclose(np.asarray(list(scene.sensors()[0].har.sample_to_camera)), np.asarray(list(ref.har.sample_to_camera)), rtol=1e-6, atol=1e-7)
    scene = mi.load_dict(cbox(32, 24, fov=35.0, fov_axis="y"))
    params = mi.traverse(scene)
    params["sensor.x_fov"] = torch.tensor([60.0]); params.update()
    ref = mi.load_dict(cbox(32, 24, fov=60.0)).sensors()[0]
    assert bytes(scene.sensors()[0].har) == bytes(ref.har)
    with pytest.raises(RuntimeError, match="field of view"):
        params["sensor.x_fov"] = torch.tensor([190.0]); params.update()
    assert bytes(scene.sensors()[0].har) == bytes(ref.har)          # a rejected value leaves the sensor as it was
    # the principal point (perspective.cpp:158-159)
    d = cbox(32, 24, fov=40.0); d["sensor"]["principal_point_offset_x"] = 0.1
    scene = mi.load_dict(d); params = mi.traverse(scene)
    assert abs(float(params["sensor.principal_point_offset_x"]) - 0.1) < 1e-7 and float(params["sensor.principal_point_offset_y"]) == 0.0
    params["sensor.principal_point_offset_x"] = torch.tensor([-0.05]); params["sensor.principal_point_offset_y"] = torch.tensor([0.2]); params.update()
    d["sensor"]["principal_point_offset_x"] = -0.05; d["sensor"]["principal_point_offset_y"] = 0.2
    assert bytes(scene.sensors()[0].har) == bytes(mi.load_dict(d).sensors()[0].har)
    assert "cam.x_fov" not in mi.traverse(mi.load_dict({"type": "scene", "cam": {"type": "orthographic", "film": {"type": "hdrfilm", "width": 8, "height": 8}}})).keys()


def test_scalar_eta_is_a_parameter_of_the_dielectric_models(mi, O):
    """'<bsdf>.eta' -- int_ior / ext_ior, a plain float -- is registered by dielectric.cpp:238, plastic.cpp:185 and roughplastic.cpp:211: readable, and written + params.update()
    the scene equals a freshly loaded one with that index (Fresnel terms, plastic's internal reflectance, roughplastic's tables are lowered from it)"""
    import torch
    def make(iors):
        d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = 20; f["height"] = 20
        d["glass"] = {"type": "dielectric", "int_ior": iors[0], "ext_ior": 1.0}
        d["coat"] = {"type": "plastic", "int_ior": iors[1], "ext_ior": 1.0, "diffuse_reflectance": {"type": "rgb", "value": [0.2, 0.4, 0.6]}}
        d["rough"] = {"type": "roughplastic", "int_ior": iors[2], "ext_ior": 1.0, "alpha": 0.2, "diffuse_reflectance": {"type": "rgb", "value": [0.5, 0.3, 0.2]}}
        d["small-box"]["bsdf"] = {"type": "ref", "id": "glass"}; d["large-box"]["bsdf"] = {"type": "ref", "id": "coat"}; d["floor"]["bsdf"] = {"type": "ref", "id": "rough"}
        return d
    scene = mi.load_dict(make([1.5, 1.49, 1.6]))
    params = mi.traverse(scene)
    assert abs(float(params["glass.eta"]) - 1.5) < 1e-6 and abs(float(params["coat.eta"]) - 1.49) < 1e-6 and abs(float(params["rough.eta"]) - 1.6) < 1e-6
    new = [1.33, 1.7, 1.45]
    for k, v in zip(("glass.eta", "coat.eta", "rough.eta"), new):
        params[k] = torch.tensor([v])
    params.update()
    want = mi.load_dict(make(new))
    o1, s1 = O.scene_from_product(scene); o2, s2 = O.scene_from_product(want)
    a, st1 = o1.render_path(s1, seed=5, spp=4, max_depth=6, threads=1); b, st2 = o2.render_path(s2, seed=5, spp=4, max_depth=6, threads=1)
    assert st1.vertices == st2.vertices and np.array_equal(a, b)
    with pytest.raises(RuntimeError, match="indices of refraction"):
        params["rough.eta"] = torch.tensor([1.0]); params.update()


def test_scene_parameters_keep_flags_set_dirty(mi):
    """the rest of SceneParameters' interface (util.py:146-256): flags(key), keep(keys) with regular expressions, set_dirty(key); update() keeps working on a reduced table"""
    import torch
    scene = mi.load_dict(mi.cornell_box())
    params = mi.traverse(scene)
    F = mi.ParamFlags
    assert params.flags("red.reflectance.value") == F.Differentiable
    assert params.flags("floor.positions") == F.Differentiable | F.Discontinuous
    assert params.flags("sensor.to_world") & F.NonDifferentiable and params.flags("sensor.film.size") & F.ReadOnly
    with pytest.raises(KeyError):
        params.flags("nonsense")
    params.keep([r".*\.reflectance\.value", "sensor.to_world"])
    assert sorted(params.keys()) == ["green.reflectance.value", "red.reflectance.value", "sensor.to_world", "white.reflectance.value"]
    params["red.reflectance.value"] = torch.tensor([0.1, 0.2, 0.3]); params.update()
    assert np.allclose(scene.bsdf_objs[scene._param_keys()["red.reflectance.value"][1].index].value, [0.1, 0.2, 0.3])
    params.keep("green.*")
    assert list(params.keys()) == ["green.reflectance.value"]
    with torch.no_grad():
        params["green.reflectance.value"].data[0] = 0.9            # a write the version counter does not see
    params.set_dirty("green.reflectance.value"); params.update()
    assert abs(float(scene.bsdf_objs[scene._param_keys()["green.reflectance.value"][1].index].value[0]) - 0.9) < 1e-7


def test_traverse_of_a_single_plugin_object(mi):
    """mi.traverse(<object>) (util.py:263-341): the reference's plugin tests traverse emitters, shapes and BSDFs on their own and use the names without a prefix -- the
    spot light's cone as in src/emitters/tests/test_spot.py:189-212, a rectangle's `to_world` as in src/shapes/tests/test_rectangle.py:234-243, a BSDF's colour"""
    import torch
    e = mi.load_dict({"type": "spot", "cutoff_angle": 20.0, "intensity": {"type": "rgb", "value": [1, 2, 3]}})
    params = mi.traverse(e)
    assert "cutoff_angle" in params and "beam_width" in params
    assert abs(float(params["cutoff_angle"]) - 20.0) < 1e-6 and abs(float(params["beam_width"]) - 15.0) < 1e-6          # beam_width defaults to 3/4 of the cutoff (spot.cpp:105-107)
    params["cutoff_angle"] = 30.0; params["beam_width"] = 20.0; params.update()
    again = mi.traverse(e)
    assert abs(float(again["cutoff_angle"]) - 30.0) < 1e-6 and abs(float(again["beam_width"]) - 20.0) < 1e-6
    with pytest.raises(RuntimeError, match="cutoff_angle"):
        again["beam_width"] = 45.0; again.update()
    r = mi.load_dict({"type": "rectangle"})
    rp = mi.traverse(r)
    assert "to_world" in rp and tuple(rp["to_world"].shape) == (4, 4)
    rp["to_world"] = torch.as_tensor(np.asarray(mi.ScalarTransform4f().scale([2.0, 2.0, 1.0]).matrix, np.float32)); rp.update()
    assert np.allclose(np.abs(mi.traverse(r)["positions"].cpu().numpy()[:, :2]), 2.0)
    b = mi.load_dict({"type": "roughplastic", "alpha": 0.2, "diffuse_reflectance": {"type": "rgb", "value": [0.1, 0.2, 0.3]}})
    bp = mi.traverse(b)
    assert sorted(bp.keys()) == ["alpha", "diffuse_reflectance.value", "eta", "specular_reflectance.value"] and abs(float(bp["alpha"]) - 0.2) < 1e-7
    bp.keep("alpha"); assert bp.keys() == ["alpha"]
    with pytest.raises(KeyError):
        bp["nonsense"] = 1.0
