"""Row a11 of SURVEY.md section 8, the spatially varying half: an `area` emitter whose `radiance` is a bitmap, on a rectangle (src/emitters/area.cpp:74,
133-165, 185-191; BitmapTexture::sample_position / pdf_position, src/textures/bitmap.cpp:622-703; DiscreteDistribution2D, include/mitsuba/core/distr_2d.h:76-180;
Rectangle::eval_parameterization, src/shapes/rectangle.cpp:215-237) on the CPU:
 * the oracle against the reference's own known answers for DiscreteDistribution2D (src/core/tests/test_distr_2d.py:168-180) and against the reference's
   chi^2 / consistency tests of sample_position (src/textures/tests/test_bitmap.py:8-29, 250-268), re-hosted with a synthetic bitmap (`carrot.png` is not in the tree);
 * the product's host-compiled shading code against the oracle, forward and prb, and against an analytic expectation;
 * the parameters (`'<shape>.emitter.radiance.data'`, `.to_uv`), and what is refused by name."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bitmap(seed=0, w=12, h=9):
    rng = np.random.default_rng(seed)
    t = rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32) ** 3        # a wide range of luminances
    t[h // 2, w // 3] = [30.0, 20.0, 10.0]                                 # one hot texel
    t[0, :2] = 0.0                                                         # and texels without any mass
    return t


def lit_box(mi, tex, res=24, **bitmap_props):
    """the Cornell box with a bitmap radiated by its ceiling light"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["light"]["emitter"] = {"type": "area", "radiance": dict({"type": "bitmap", "data": tex}, **bitmap_props)}
    return d


def _oracle_lib(O):
    return O.lib()


def test_discrete_distribution_2d_known_answers(O):
    """the six known answers of the reference's test05_discrete_distribution_2d (src/core/tests/test_distr_2d.py:168-180)"""
    L = _oracle_lib(O)
    vals = np.array([[1, 2, 3], [0, 1, 3]], np.float32)
    pts = np.array([[0, 0], [1.0 / 6.0 - 1e-7, 0], [1.0 / 6.0 + 1e-7, 0], [1, 0], [0, 6 / 10 - 1e-7], [0, 6 / 10 + 1e-7]], np.float32)
    n = len(pts); pos = np.zeros((n, 2), np.uint32); pmf = np.zeros(n, np.float32); re = np.zeros((n, 2), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.orc_discrete_distribution_2d_sample(vp(vals), 3, 2, vp(pts), n, vp(pos), vp(pmf), vp(re)) == 0
    want = [([0, 0], .1, [0, 0]), ([0, 0], .1, [1, 0]), ([1, 0], .2, [0, 0]), ([2, 0], .3, [1, 0]), ([0, 0], .1, [0, 1]), ([1, 1], .1, [0, 0])]
    for i, (p, m, r) in enumerate(want):
        assert list(pos[i]) == p and abs(pmf[i] - m) < 1e-6 and np.allclose(re[i], r, atol=1e-6), (i, pos[i], pmf[i], re[i])


def _texture_functions(mi, O, tex, **bitmap_props):
    scene = mi.load_dict(lit_box(mi, tex, 8, **bitmap_props))
    from tests.test_cpu_host import oracle_scene_from
    osc, _ = oracle_scene_from(O, scene)
    L = _oracle_lib(O)
    index = [i for i, e in enumerate(scene.emitters) if e["type"] == 7][0]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def sample(s):
        s = np.ascontiguousarray(s, np.float32); n = len(s); uv = np.zeros((n, 2), np.float32); pdf = np.zeros(n, np.float32)
        assert L.orc_emitter_texture_sample_position(osc.handle, index, vp(s), n, vp(uv), vp(pdf), 0) == 0
        return uv, pdf

    def pdf(p):
        p = np.ascontiguousarray(p, np.float32); n = len(p); out = np.zeros(n, np.float32)
        assert L.orc_emitter_texture_sample_position(osc.handle, index, vp(p), n, None, vp(out), 1) == 0
        return out
    return sample, pdf, osc


@pytest.mark.parametrize("filter_type", ["nearest", "bilinear"])
@pytest.mark.parametrize("wrap_mode", ["repeat", "clamp", "mirror"])
def test_sample_position_consistency_and_histogram(mi, O, filter_type, wrap_mode):
    """the reference's test07_sample_position_consistency (pdf returned by sample_position == pdf_position at the sampled point, test_bitmap.py:250-268) and its
    test01 (the samples are distributed like pdf_position, :8-29; here as a histogram test on a 48 x 36 grid with 1.5M stratified samples)"""
    tex = _bitmap(3)
    sample, pdf, _ = _texture_functions(mi, O, tex, filter_type=filter_type, wrap_mode=wrap_mode)
    n = 400
    g = (np.arange(n, dtype=np.float32)) / (n - 1)
    s = np.stack(np.meshgrid(g, g), -1).reshape(-1, 2)
    uv, p = sample(s)
    assert np.allclose(p, pdf(uv), rtol=1e-5, atol=1e-6)
    assert (uv >= 0).all() and (uv <= 1).all() and np.isfinite(p).all() and (p >= 0).all()
    # histogram of 1.5M independent samples against the integral of pdf_position over each cell (midpoint rule on a 6 x 6 sub-grid): Poisson statistics per cell
    rng = np.random.default_rng(11)
    uv, _ = sample(rng.random((1500000, 2), dtype=np.float32))
    bx, by = 48, 36
    hist, _, _ = np.histogram2d(uv[:, 1], uv[:, 0], bins=[by, bx], range=[[0, 1], [0, 1]])
    sub = 6
    xs = (np.arange(bx * sub) + 0.5) / (bx * sub); ys = (np.arange(by * sub) + 0.5) / (by * sub)
    dens = pdf(np.stack(np.meshgrid(xs, ys), -1).reshape(-1, 2).astype(np.float32)).reshape(by * sub, bx * sub)
    expect = dens.reshape(by, sub, bx, sub).mean(axis=(1, 3)) / (bx * by) * len(uv)
    assert abs(expect.sum() / len(uv) - 1.0) < 2e-2                       # the density integrates to one
    big = expect > 30
    z = (hist[big] - expect[big]) / np.sqrt(expect[big])
    assert big.sum() > 0.8 * big.size and np.abs(z).max() < 6.0 and 0.7 < (z ** 2).mean() < 1.4, (np.abs(z).max(), (z ** 2).mean())
    assert hist[expect == 0].sum() == 0                                   # texels without mass are never sampled (nearest) / only through their neighbours' tents


def test_host_pipeline_matches_oracle_and_expectation(mi, O):
    """the product's shading code (compiled for the host) == the oracle, forward and prb primal; and a plausibility check that needs no reference: with nearest
    filtering the light's irradiance on the floor is that of a uniform light of the bitmap's mean radiance wherever the light is far enough to look uniform --
    tested in the weak form 'the image mean scales with the bitmap' (linearity in the radiance)"""
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    tex = _bitmap(5)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    images = {}
    for props in ({}, {"filter_type": "nearest", "wrap_mode": "clamp"}, {"wrap_mode": "mirror", "to_uv": mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]])}):
        scene = mi.load_dict(lit_box(mi, tex, 24, **props))
        assert [e["type"] for e in scene.emitters] == [7]
        osc, sensor = oracle_scene_from(O, scene)
        desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
        for mode, md in ((0, 6), (1, 5)):
            film = np.zeros((24, 24, 4), np.float32)
            assert L.hh_render(h, C.byref(sensor), mode, 3, 16, md, 5, 0, 0, O.fp(film)) == 0
            ref, st = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=16, max_depth=md, raw=True, threads=2)
            assert np.isfinite(film).all() and film[..., :3].max() > 0 and rel_l2(O.develop(film), O.develop(ref)) < 1e-4, (props, mode)
            images[(len(images))] = O.develop(ref)
    # linearity: twice the bitmap, twice the picture (same sample stream: the distribution is scale-free)
    scene2 = mi.load_dict(lit_box(mi, 2.0 * tex, 24)); osc2, sensor = oracle_scene_from(O, scene2)
    ref2, _ = osc2.render_path(sensor, seed=3, spp=16, max_depth=6, raw=True, threads=2)
    assert rel_l2(O.develop(ref2), 2.0 * images[0]) < 1e-5


def test_uniform_bitmap_equals_uniform_light_in_expectation(mi, O):
    """a constant bitmap is a uniform light: not the same estimator (texel choice + tent instead of a uniform point on the rectangle), the same integral"""
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    rad = [18.387, 13.9873, 6.75357]
    tex = np.broadcast_to(np.asarray(rad, np.float32), (4, 4, 3)).copy()
    a_scene = mi.load_dict(lit_box(mi, tex, 16)); b = mi.cornell_box(); b["sensor"]["film"]["width"] = 16; b["sensor"]["film"]["height"] = 16
    b_scene = mi.load_dict(b)
    osc_a, sensor = oracle_scene_from(O, a_scene); osc_b, _ = oracle_scene_from(O, b_scene)
    ia, _ = osc_a.render_path(sensor, seed=1, spp=2048, max_depth=4, raw=True, threads=8); ib, _ = osc_b.render_path(sensor, seed=2, spp=2048, max_depth=4, raw=True, threads=8)
    a, b = O.develop(ia), O.develop(ib)
    assert rel_l2(a, b) < 0.04                                                              # two independent 2048-spp renders of the uniform light differ by 0.025
    assert np.abs(a.mean(axis=(0, 1)) / b.mean(axis=(0, 1)) - 1.0).max() < 0.02           # the means agree (two uniform renders: 1.2 %)
    # 4 x 4 blocks of pixels
    ab, bb = a.reshape(4, 4, 4, 4, 3).mean(axis=(1, 3)), b.reshape(4, 4, 4, 4, 3).mean(axis=(1, 3))
    assert np.abs(ab / bb - 1.0).max() < 0.07


@pytest.mark.parametrize("props", [{}, {"filter_type": "nearest"}, {"wrap_mode": "mirror", "to_uv": "swap"}])
def test_oracle_light_texel_gradients(mi, O, props):
    """`radiance` of an area light is a differentiable traverse entry (area.cpp:64-70): the oracle's gradient w.r.t. the bitmap's texels -- emission term at si.uv (prb.py:160-161 with
    emitter.eval attached), emitter-sampling term at the SAMPLED ds.uv with the sampling density detached (prb.py:174-175, 203-206).
     * exact: the image is homogeneous of degree one in the texels and scaling them leaves the texel distribution alone, so  sum_j grad_j * texel_j == sum(w * image)  for the
       same seed (the light is the scene's only emitter);
     * finite differences of the oracle's own primal renders along a random direction in texel space (the samples follow the changed distribution there: two estimators of one
       derivative, 4096 spp, 3 %)"""
    from tests.test_cpu_host import oracle_scene_from
    props = dict(props)
    if props.get("to_uv") == "swap":
        props["to_uv"] = mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]])
    res = 12
    tex = _bitmap(5, 6, 5)
    tex[0, :2] = 0.02               # (keep every texel sampled: a texel without mass has a gradient that no finite difference of ITS value can see through the sampled term ... it can: the hit term)
    scene = mi.load_dict(lit_box(mi, tex, res, **props))
    osc, sensor = oracle_scene_from(O, scene)
    ti = [k for k, t in enumerate(osc.data.textures) if t.shape == tex.shape][-1]
    w = np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    kw = dict(seed=9, spp=4096, max_depth=3)
    _, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, w, **kw)
    g = g_tex[ti].astype(np.float64)
    assert np.abs(g).max() > 0 and not g_emit.any()
    img, _ = osc.render_prb(sensor, **kw)
    total = float((img.astype(np.float64) * w).sum())
    assert abs(float((g * tex).sum()) - total) < 2e-4 * abs(total), (float((g * tex).sum()), total)
    direction = np.random.default_rng(3).uniform(-1.0, 1.0, tex.shape) * tex          # relative perturbations: texels stay non-negative
    sums = []
    for sgn in (+1, -1):
        osc.set_texture(ti, (tex + sgn * 0.02 * direction).astype(np.float32))
        im, _ = osc.render_prb(sensor, **kw)
        sums.append(float((im.astype(np.float64) * w).sum()))
    osc.set_texture(ti, tex)
    fd = (sums[0] - sums[1]) / 0.04
    ad = float((g * direction).sum())
    assert abs(fd - ad) < 0.03 * abs(total) * 0.5 + 0.03 * abs(fd), (fd, ad, total)


@pytest.mark.parametrize("props", [{}, {"filter_type": "nearest", "wrap_mode": "clamp"}, {"wrap_mode": "mirror", "to_uv": "swap"}, {"zeros": True}])
def test_host_light_texel_adjoint_matches_oracle(mi, O, props):
    """what k_shade commits under HAR_SHADE_LIGHT_TEXELS (shade_lane's lt_* fields, em_unit / contrib_unit, the sample's visibility), run lane by lane on the host, against the
    oracle, texel by texel; `zeros`: a bitmap with a block of zero texels -- a sample over them contributes nothing and still has a derivative (its shadow ray is traced)"""
    from tests.test_cpu_host import oracle_scene_from
    props = dict(props)
    if props.get("to_uv") == "swap":
        props["to_uv"] = mi.ScalarTransform3f([[0, 1, 0], [1, 0, 0], [0, 0, 1]])
    tex = _bitmap(5, 6, 5)
    if props.pop("zeros", False):
        tex[:3, :3] = 0.0
    res = 16
    scene = mi.load_dict(lit_box(mi, tex, res, **props))
    osc, sensor = oracle_scene_from(O, scene)
    ti = scene.emitters[0]["light"].tex_index
    w = np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    kw = dict(seed=3, spp=16, max_depth=5)
    _, g_tex, _, _ = osc.render_prb_backward_emitters(sensor, w, **kw)
    want = g_tex[ti].astype(np.float64)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    dp = C.POINTER(C.c_double)
    L.hh_render_backward_light_texels.argtypes = [C.c_void_p, C.c_void_p, O.c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(dp)]
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((res, res, 4), np.float32)
    assert L.hh_render(h, C.byref(sensor), 1, kw["seed"], kw["spp"], kw["max_depth"], 5, 0, 0, O.fp(film)) == 0
    wt = film[:, :, 3:4]; adj = np.ascontiguousarray(w / np.where(wt == 0, 1, wt), np.float32)
    got = np.zeros(tex.shape, np.float64)
    nt = len(scene.textures)
    pp = (dp * nt)(*[got.ctypes.data_as(dp) if k == ti else dp() for k in range(nt)])
    assert L.hh_render_backward_light_texels(h, C.byref(sensor), O.fp(adj), kw["seed"], kw["spp"], kw["max_depth"], 5, pp) == 0
    scale = np.abs(want).max()
    assert scale > 0 and np.abs(got - want).max() < 1e-3 * scale, np.abs(got - want).max() / scale
    if "zeros" in str(props) or not tex[:3, :3].any():
        assert np.abs(want[:2, :2]).max() > 0          # the zero block does receive a gradient


def test_parameters_and_refusals(mi, O):
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    tex = _bitmap(7)
    scene = mi.load_dict(lit_box(mi, tex, 12))
    params = mi.traverse(scene)
    assert "light.emitter.radiance.data" in params and "light.emitter.radiance.to_uv" in params and "light.emitter.sampling_weight" in params
    assert "light.emitter.radiance.value" not in params
    assert tuple(params["light.emitter.radiance.data"].shape) == tex.shape
    # new texels: the scene equals a freshly loaded one (records and oracle renders)
    import torch
    tex2 = _bitmap(8)
    params["light.emitter.radiance.data"] = torch.tensor(tex2); params.update()
    fresh = mi.load_dict(lit_box(mi, tex2, 12))
    oa, sensor = oracle_scene_from(O, scene); ob, _ = oracle_scene_from(O, fresh)
    ia, _ = oa.render_path(sensor, seed=4, spp=8, max_depth=5, raw=True, threads=2); ib, _ = ob.render_path(sensor, seed=4, spp=8, max_depth=5, raw=True, threads=2)
    assert np.array_equal(ia, ib)
    # a flip of the texture is a to_uv that keeps the unit square; a rotation by 30 degrees is not (bitmap.cpp:976-992)
    params["light.emitter.radiance.to_uv"] = torch.tensor(mi.ScalarTransform3f([[-1, 0, 1], [0, 1, 0], [0, 0, 1]]).matrix); params.update()
    with pytest.raises(RuntimeError, match="maps the unit square onto"):
        params["light.emitter.radiance.to_uv"] = torch.tensor(mi.ScalarTransform3f().rotate(30.0).matrix); params.update()
    with pytest.raises(RuntimeError, match="maps the unit square onto"):
        mi.load_dict(lit_box(mi, tex, 12, to_uv=mi.ScalarTransform3f().scale([2.0, 1.0])))
    with pytest.raises(RuntimeError, match="non-negative"):
        params["light.emitter.radiance.data"] = torch.tensor(-tex2); params.update()
    # on a triangle mesh the reference needs Mesh::eval_parameterization: refused by name
    d = lit_box(mi, tex, 12); d["small-box"]["emitter"] = {"type": "area", "radiance": {"type": "bitmap", "data": tex}}
    with pytest.raises(RuntimeError, match="eval_parameterization"):
        mi.load_dict(d)
    # other textured emissive parameters stay refused
    d = lit_box(mi, tex, 12); d["env"] = {"type": "constant", "radiance": {"type": "bitmap", "data": tex}}
    with pytest.raises(RuntimeError, match="spatially varying"):
        mi.load_dict(d)


def test_xml_scene_with_a_bitmap_radiance(mi, tmp_path):
    """<emitter type="area"><texture type="bitmap" name="radiance"> ... : the nested texture reaches the emitter through the XML parser like it does through load_dict"""
    t = _bitmap(5)
    path = tmp_path / "light.pfm"
    mi.Bitmap(t).write(str(path))
    xml = ('<scene version="3.0.0">'
           '<shape type="rectangle" id="lamp"><emitter type="area"><texture type="bitmap" name="radiance"><string name="filename" value="%s"/>'
           '<string name="wrap_mode" value="clamp"/><string name="filter_type" value="nearest"/></texture><float name="sampling_weight" value="2"/></emitter></shape>'
           '<shape type="rectangle" id="floor"><transform name="to_world"><translate z="-1"/></transform></shape></scene>') % path
    sc = mi.load_string(xml)
    assert [e["type"] for e in sc.emitters] == [7] and sc.emitters[0]["sampling_weight"] == 2.0
    assert sc.texture_modes[sc.emitters[0]["radiance_texture"]] == 5 and np.array_equal(sc.textures[sc.emitters[0]["radiance_texture"]], t)
