"""GPU parity tests (run on the MI355X box): HIP path vs the CPU oracle through the C ABI.

Tolerances are north_star's: forward images 1e-4, PRB gradients 1e-3 relative L2; ray
queries (integer / index work and the exact Moeller-Trumbore arithmetic) bit-exact.
Modelled on the reference's src/render/tests/test_kdtrees.py (accel == brute force),
src/integrators/tests/test_integrators.py:28-53 and src/render/tests/test_ad.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def cbox(mi, O, res, crop=None, rfilter="gaussian"):
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["sensor"]["film"]["rfilter"] = {"type": rfilter}
    if crop:
        f = d["sensor"]["film"]
        f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = crop
    sd, sensor = O.cornell_box(res, res, crop=crop, rfilter=rfilter)
    return mi.load_dict(d), O.OracleScene(sd), sensor


def random_rays(n, seed=1):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-1, 1, (3, n)).astype(np.float32)
    d = rng.normal(size=(3, n)).astype(np.float32); d /= np.linalg.norm(d, axis=0)
    return o, d.astype(np.float32)


def test_device_is_gfx950(mi):
    arch = mi.lib().har_device_arch()
    assert arch is not None and arch.decode().startswith("gfx950")


def test_ray_intersect_bitexact_cornell(mi, O):
    scene, osc, _ = cbox(mi, O, 32)
    n = 300000
    o, d = random_rays(n)
    maxt = np.full(n, 3.402823466e+38, np.float32)
    ref = osc.ray_intersect(o, d, maxt, naive=True)
    for naive in (False, True):
        pi = scene._intersect(mi.Ray3f(o, d, maxt), naive)
        t = pi.t.cpu().numpy(); hit = np.isfinite(ref[0])
        assert np.array_equal(t, ref[0])
        assert np.array_equal(pi.prim_uv[0].cpu().numpy()[hit], ref[1][hit])
        assert np.array_equal(pi.prim_uv[1].cpu().numpy()[hit], ref[2][hit])
        assert np.array_equal(pi.prim_index.cpu().numpy().astype(np.uint32)[hit], ref[3][hit])
        assert np.array_equal(pi.shape_index.cpu().numpy().astype(np.uint32)[hit], ref[4][hit])
    maxt2 = np.random.default_rng(2).uniform(0.05, 2.5, n).astype(np.float32)
    assert np.array_equal(scene.ray_test(mi.Ray3f(o, d, maxt2)).cpu().numpy(), osc.ray_test(o, d, maxt2))


def test_path_directly_visible_kat(mi, O):
    """test_integrators.py:28-53: pixel (124,36) of the Cornell box, max_depth=1 sees only the emitter."""
    scene, _, _ = cbox(mi, O, 256, crop=(124, 36, 1, 1))
    integ = mi.load_dict({"type": "path", "max_depth": 1})
    img = mi.render(scene, integrator=integ, spp=64).cpu().numpy()
    assert np.allclose(img.reshape(3), [18.387, 13.9873, 6.75357], rtol=1e-5)


@pytest.mark.parametrize("res,spp,seed", [(64, 16, 0), (96, 4, 3), (33, 7, 1)])
def test_forward_path_parity(mi, O, res, spp, seed):
    scene, osc, sensor = cbox(mi, O, res)
    img = mi.render(scene, spp=spp, seed=seed).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=seed, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    gst = scene.integrator().stats()
    assert gst["paths"] == st.paths and gst["vertices"] == st.vertices


def test_forward_box_filter_and_crop(mi, O):
    scene, osc, sensor = cbox(mi, O, 64, crop=(8, 16, 40, 24), rfilter="box")
    img = mi.render(scene, spp=8, seed=5).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=5, spp=8, max_depth=8)
    assert img.shape == (24, 40, 3)
    assert rel_l2(img, ref) < 1e-4


def test_forward_chunked_equals_single(mi, O):
    """Several wavefront chunks and lane sub-ranges reproduce the single-launch film."""
    scene, osc, sensor = cbox(mi, O, 64)
    small = mi.load_dict({"type": "path", "max_depth": 8, "chunk_lanes": 4096})
    a = mi.render(scene, integrator=small, spp=8, seed=2).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=2, spp=8, max_depth=8)
    assert rel_l2(a, ref) < 1e-4
    total = 64 * 64 * 8
    f = small.render_film(scene, 0, 2, 8, lanes=(0, total // 3))
    small.render_film(scene, 0, 2, 8, lanes=(total // 3, total), film=f)
    assert rel_l2(mi.develop_film(f).cpu().numpy(), ref) < 1e-4


def test_prb_primal_parity(mi, O):
    scene, osc, sensor = cbox(mi, O, 64)
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=16, seed=0).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=0, spp=16, max_depth=6)
    assert rel_l2(img, ref) < 1e-4


def _textured(mi, O, res, tex_res, spp):
    d = mi.textured_cornell_box(res=res, tex_res=tex_res, spp=spp)
    scene = mi.load_dict(d)
    tex = d["white"]["reflectance"]["data"]
    sd, sensor = O.cornell_box(res, res, white_texture=tex)
    return scene, O.OracleScene(sd), sensor


def test_prb_backward_texture_gradient(mi, O):
    """C4 at test size: albedo-texture gradient of loss = mean(img^2)."""
    import torch
    res, spp = 48, 16
    scene, osc, sensor = _textured(mi, O, res, 16, spp)
    params = mi.traverse(scene)
    key = "white.reflectance.data"
    params[key].requires_grad_()
    img = mi.render(scene, params, spp=spp, seed=0)
    (img ** 2).mean().backward()
    g = params[key].grad.cpu().numpy()
    ref_img, _ = osc.render_prb(sensor, seed=0, spp=spp, max_depth=6)
    assert rel_l2(img.detach().cpu().numpy(), ref_img) < 1e-4
    grad_in = 2.0 * ref_img / ref_img.size
    seed_grad = mi.sample_tea_32(0, 1)[0]
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=seed_grad, spp=spp, max_depth=6)
    assert rel_l2(g, g_tex[0]) < 1e-3


def test_texel_gradient_fixed_point_accumulation_edge_cases(mi, O):
    """k_texel_accumulate adds the queued texel gradients in 64-bit fixed point scaled by the launch's largest gradient component: (a) an adjoint image whose
    pixels span 24 orders of magnitude still matches the oracle's float sums to 1e-3; (b) a
    tiny and a huge uniform scale give the same gradient up to that scale (no overflow, no underflow); (c) an infinite pixel adjoint takes the float fallback
    and reaches the texture as a non-finite value instead of being dropped"""
    res, spp = 48, 16
    scene, osc, sensor = _textured(mi, O, res, 16, spp)
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    rng = np.random.default_rng(5)
    base = rng.uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    key = "white.reflectance.data"
    g1 = integ.render_backward(scene, None, base, seed=11, spp=spp)[key].cpu().numpy()
    _, ref, _ = osc.render_prb_backward(sensor, base, seed=11, spp=spp, max_depth=6)
    assert rel_l2(g1, ref[0]) < 1e-3
    for scale in (1e-30, 1e+30):
        gs = integ.render_backward(scene, None, (base * np.float32(scale)).astype(np.float32), seed=11, spp=spp)[key].cpu().numpy()
        assert np.isfinite(gs).all() and rel_l2(gs / np.float32(scale), g1) < 1e-5, scale
    wide = base.copy(); wide[: res // 2] *= np.float32(1e-12); wide[res // 2:] *= np.float32(1e+12)      # two halves of the film, 24 decades apart
    gw = integ.render_backward(scene, None, wide, seed=11, spp=spp)[key].cpu().numpy()
    _, rw, _ = osc.render_prb_backward(sensor, wide, seed=11, spp=spp, max_depth=6)
    assert rel_l2(gw, rw[0]) < 1e-3
    bad = base.copy(); bad[res // 2, res // 2, 0] = np.inf
    gb = integ.render_backward(scene, None, bad, seed=11, spp=spp)[key].cpu().numpy()
    assert not np.isfinite(gb).all()


def test_prb_backward_constant_albedo(mi, O):
    scene, osc, sensor = cbox(mi, O, 40)
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    grad_in = np.random.default_rng(0).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
    grads = integ.render_backward(scene, None, grad_in, seed=11, spp=8)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=11, spp=8, max_depth=6)
    got = np.stack([grads[k].cpu().numpy() for k in ("white.reflectance.value", "green.reflectance.value", "red.reflectance.value")])
    assert rel_l2(got, g_refl) < 1e-3


def test_instanced_scene_parity(mi, O):
    """TLAS/BLAS path (src/shapes/instance.cpp): small instanced scene, rays + image vs oracle."""
    d = mi.instanced_spheres_scene(width=48, height=48, spp=8, grid=3, n_u=12, n_v=6)
    scene = mi.load_dict(d)
    from tests.test_cpu_host import oracle_scene_from
    osc, sensor = oracle_scene_from(O, scene)
    n = 100000
    o, dd = random_rays(n, 7)
    maxt = np.full(n, 3.402823466e+38, np.float32)
    ref = osc.ray_intersect(o, dd, maxt, naive=True)
    pi = scene.ray_intersect_preliminary(mi.Ray3f(o, dd, maxt))
    hit = np.isfinite(ref[0])
    assert np.array_equal(pi.t.cpu().numpy(), ref[0])
    assert np.array_equal(pi.instance.cpu().numpy().astype(np.uint32)[hit], ref[5][hit])
    img = mi.render(scene, spp=8, seed=1).cpu().numpy()
    ref_img, _ = osc.render_path(sensor, seed=1, spp=8, max_depth=8)
    assert rel_l2(img, ref_img) < 1e-4


def test_sampler_matches_oracle_stream(mi, O):
    import ctypes as C
    s = mi.Sampler({"sample_count": 4})
    s.seed(7, 1000)
    a = s.next_1d().cpu().numpy(); b = s.next_2d().cpu().numpy()
    for lane in (0, 1, 513, 999):
        out = np.empty(3, np.float32)
        O.lib().orc_sampler_stream(7, lane, 3, O.fp(out))
        assert a[lane] == out[0] and b[0, lane] == out[1] and b[1, lane] == out[2]


def test_bsdf_diffuse_closed_form(mi):
    """src/bsdfs/tests/test_diffuse.py:16-39"""
    import torch
    bsdf = mi.load_dict({"type": "diffuse"})
    si = type("SI", (), dict(wi=torch.tensor([0.0, 0.0, 1.0]), uv=None))()
    theta = np.arange(20) / 19.0 * (np.pi / 2)
    wo = np.stack([np.sin(theta), np.zeros(20), np.cos(theta)]).astype(np.float32)
    val, pdf = bsdf.eval_pdf(mi.BSDFContext(), si, wo)
    assert np.allclose(pdf.cpu().numpy(), wo[2] / np.pi, atol=1e-6)
    assert np.allclose(val.cpu().numpy()[0], 0.5 * wo[2] / np.pi, atol=1e-6)


def test_full_size_properties(mi):
    """BASELINE config 2 size (512 x 512 x 256 spp): size-independent checks -- weight channel ==
    spp-normalised constant, image finite and non-negative, energy matches a low-spp render."""
    import torch
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
    scene = mi.load_dict(d)
    integ = scene.integrator()
    film = integ.render_film(scene, 0, 0, 256)
    torch.cuda.synchronize()
    st = integ.stats()
    assert st["paths"] == 512 * 512 * 256
    w = film[..., 3]
    # un-normalised Gaussian (sigma = 0.5): sum of the 5x5 weights ~ 2*pi*sigma^2 = 1.5708 per sample, minus what leaves the film
    assert abs(float(w.sum()) / (512 * 512 * 256) / (2 * np.pi * 0.25) - 1.0) < 2e-2
    img = mi.develop_film(film)
    assert bool(torch.isfinite(img).all()) and float(img.min()) >= 0.0
    low = mi.render(scene, spp=16, seed=9)
    assert abs(float(img.mean()) / float(low.mean()) - 1.0) < 2e-2


# ---------------------------------------------------------------- committed golden fixtures (tests/golden/)

def _fx():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_fixtures.npz"))


def test_golden_ray_queries_bitexact(mi):
    """HIP BVH8 traversal vs the committed brute-force oracle results: t/u/v/prim/shape bit for bit."""
    fx = _fx()
    scene = mi.load_dict(mi.cornell_box())
    n = fx["rays_o"].shape[1]
    ray = mi.Ray3f(fx["rays_o"], fx["rays_d"], np.full(n, 3.402823466e+38, np.float32))
    for pi in (scene.ray_intersect_preliminary(ray), scene._intersect(ray, True)):
        assert np.array_equal(pi.t.cpu().numpy(), fx["cornell_hit_t"])
        hit = np.isfinite(fx["cornell_hit_t"])
        assert np.array_equal(pi.prim_uv[0].cpu().numpy()[hit], fx["cornell_hit_u"][hit])
        assert np.array_equal(pi.prim_uv[1].cpu().numpy()[hit], fx["cornell_hit_v"][hit])
        assert np.array_equal(pi.prim_index.cpu().numpy().astype(np.uint32)[hit], fx["cornell_hit_prim"][hit])
        assert np.array_equal(pi.shape_index.cpu().numpy().astype(np.uint32)[hit], fx["cornell_hit_shape"][hit])
    occl = scene.ray_test(mi.Ray3f(fx["rays_o"], fx["rays_d"], np.full(n, 1.0, np.float32)))
    assert np.array_equal(occl.cpu().numpy().astype(bool), fx["cornell_ray_test_maxt1"].astype(bool))


def test_golden_forward_images(mi):
    fx = _fx()
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    scene = mi.load_dict(d)
    img = mi.render(scene, spp=8, seed=0).cpu().numpy()
    assert rel_l2(img, fx["cornell32_spp8_seed0_path"]) < 1e-4            # north_star forward tolerance
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=8, seed=0).cpu().numpy()
    assert rel_l2(img, fx["cornell32_spp8_seed0_prb"]) < 1e-4


def test_golden_prb_texture_gradient(mi):
    fx = _fx()
    d = mi.textured_cornell_box(res=24, tex_res=8, spp=8)
    d["white"]["reflectance"]["data"] = fx["c4_texture"]
    scene = mi.load_dict(d)
    integ = scene.integrator()
    grads = integ.render_backward(scene, None, fx["c4_grad_in"], seed=0x1234, spp=8)
    g = grads["white.reflectance.data"].cpu().numpy()
    assert rel_l2(g, fx["c4_grad_tex"]) < 1e-3                             # north_star PRB tolerance


def test_golden_shape_gradients(mi):
    """vertex-position gradients of prb against the committed oracle fixture (no oracle call at run time)"""
    from tests.test_shape_gradients_cpu import slab_scene
    fx = _fx()
    d = slab_scene(mi, 16, textured=True)
    d["integrator"] = {"type": "prb", "max_depth": 5, "shape_gradients": True}
    scene = mi.load_dict(d)
    grads = scene.integrator().render_backward(scene, None, fx["shape_grad_in"], seed=3, spp=16)
    for name in ("floor", "ceiling"):
        got = grads[name + ".positions"].cpu().numpy().reshape(-1, 3); want = fx["shape_grad_" + name]
        assert np.abs(got - want).max() < 2e-3 * np.abs(want).max(), name


def test_golden_round4_plugins(mi):
    """`point` / `spot` / `directional` emitters and the `orthographic` sensor in one scene against the committed oracle arrays (no oracle call at run time):
    forward image, prb image, gradients w.r.t. the four emitters' parameters and the constant albedos"""
    from tests.test_golden_cpu import round4_scene
    fx = _fx()
    scene = mi.load_dict(round4_scene(mi))
    img = mi.render(scene, spp=8, seed=2).cpu().numpy()
    assert rel_l2(img, fx["r4_path"]) < 1e-4                               # north_star forward tolerance
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=8, seed=2).cpu().numpy()
    assert rel_l2(img, fx["r4_prb"]) < 1e-4
    grads = integ.render_backward(scene, None, fx["r4_grad_in"], seed=5, spp=8)
    keys = scene._param_keys()
    ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
    assert len(ek) == 4
    got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([fx["r4_grad_emit"][i] for i in ek.values()])
    assert rel_l2(got, want) < 1e-3                                        # north_star PRB tolerance
    for k, (kind, b) in keys.items():
        if kind == "rgb":
            assert rel_l2(grads[k].cpu().numpy(), fx["r4_grad_refl"][b.index]) < 1e-3, k


def test_golden_sampler_streams(mi):
    fx = _fx()
    s = mi.Sampler({"sample_count": 4}); s.seed(7, 16)
    got = np.stack([s.next_1d().cpu().numpy() for _ in range(5)], axis=1)
    assert np.array_equal(got, fx["sampler_seed7"])


# ---------------------------------------------------------------- edge cases and size-independent properties

def test_lane_bands_union_equals_whole(mi, O):
    """rendering lane bands [0,a), [a,b), [b,N) (whole pixel rows and ragged cuts) into one film equals
    the single-call render (same per-lane streams; only the atomic summation order differs)."""
    import torch
    scene, _, _ = cbox(mi, O, 40)
    integ = scene.integrator()
    spp = 8; n = 40 * 40 * spp
    whole = integ.render_film(scene, 0, 5, spp)
    film = None
    for lo, hi in ((0, 13 * 40 * spp), (13 * 40 * spp, 13 * 40 * spp + 777), (13 * 40 * spp + 777, n)):
        film = integ.render_film(scene, 0, 5, spp, lanes=(lo, hi), film=film)
    torch.cuda.synchronize()
    assert rel_l2(film.cpu().numpy(), whole.cpu().numpy()) < 1e-6


def test_depth_limits_and_rr(mi, O):
    """max_depth 0 (weights only, path.cpp:102), 1 (emitters only), 2, and rr_depth = 1 against the oracle."""
    scene, osc, sensor = cbox(mi, O, 32)
    for md, rr in ((0, 5), (1, 5), (2, 5), (8, 1), (-1, 3)):
        integ = mi.load_dict({"type": "path", "max_depth": md, "rr_depth": rr})
        img = mi.render(scene, integrator=integ, spp=8, seed=2).cpu().numpy()
        ref, _ = osc.render_path(sensor, seed=2, spp=8, max_depth=md, rr_depth=rr)
        if md == 0:
            assert not img.any() and not ref.any()
        else:
            assert rel_l2(img, ref) < 1e-4, (md, rr)


def test_scene_without_emitters_and_all_miss(mi, O):
    """n_emitters == 0 (scene.cpp:139 pmf guard) and a sensor that sees nothing: black images, no NaNs, finite stats."""
    import torch
    d = mi.cornell_box(); d.pop("light")
    d["sensor"]["film"]["width"] = 24; d["sensor"]["film"]["height"] = 24
    img = mi.render(mi.load_dict(d), spp=4, seed=0)
    assert bool(torch.isfinite(img).all()) and float(img.abs().max()) == 0.0
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = 16; d["sensor"]["film"]["height"] = 16
    d["sensor"]["to_world"] = mi.ScalarTransform4f().look_at(origin=[0, 0, 3.9], target=[0, 0, 10], up=[0, 1, 0])
    scene = mi.load_dict(d)
    img = mi.render(scene, spp=4, seed=0)
    assert float(img.abs().max()) == 0.0
    st = scene.integrator().stats()
    assert st["paths"] == 16 * 16 * 4 and st["shadow_rays"] == 0


def test_single_pixel_film_and_odd_spp(mi, O):
    sd, sensor = O.cornell_box(1, 1)
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 1; d["sensor"]["film"]["height"] = 1
    img = mi.render(mi.load_dict(d), spp=37, seed=4).cpu().numpy()
    ref, _ = O.OracleScene(sd).render_path(sensor, seed=4, spp=37, max_depth=8)
    assert rel_l2(img, ref) < 1e-4


def test_render_refuses_more_than_2_32_lanes(mi):
    """integrator.cpp:276-294: a `path` job of 2^34 samples is split into passes of 1024 / 5 = 204 samples, which the sampler
    refuses (1024 % 204 != 0, sampler.cpp:93-94); common.py:358-363: the AD integrators refuse outright."""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 4096; d["sensor"]["film"]["height"] = 4096
    scene = mi.load_dict(d)
    with pytest.raises(Exception, match="multiple of samples_per_wavefront"):
        mi.render(scene, spp=1024)
    with pytest.raises(Exception, match="2\\^32"):
        mi.render(scene, integrator=mi.load_dict({"type": "prb", "max_depth": 6}), spp=1024)
    assert mi.load_dict({"type": "path", "max_depth": 8, "samples_per_pass": 128}).pass_layout(scene.sensors()[0], 1024) == (128, 8)


def test_multipass_render_parity(mi, O):
    """Multi-pass `path` render (integrator.cpp:173-183,276-356; SURVEY 8e: C5 runs as 8 passes of 128 spp): every lane keeps
    its pixel and continues its sampler stream from pass to pass.  Compared with the oracle's multi-pass driver, with chunks
    smaller than a pass, lane bands, max_depth 0 and the box filter."""
    import torch
    scene, osc, sensor = cbox(mi, O, 48)
    for spp, per_pass, chunk in ((16, 4, 0), (12, 3, 4096), (8, 8, 0)):
        integ = mi.load_dict({"type": "path", "max_depth": 8, "samples_per_pass": per_pass, "chunk_lanes": chunk})
        img = mi.render(scene, integrator=integ, spp=spp, seed=2).cpu().numpy()
        ref, st = osc.render_path_passes(sensor, seed=2, spp=spp, spp_per_pass=per_pass, max_depth=8)
        assert rel_l2(img, ref) < 1e-4, (spp, per_pass, chunk)
        gst = integ.stats()
        assert gst["paths"] == st.paths and gst["vertices"] == st.vertices
    single, _ = osc.render_path(sensor, seed=2, spp=16, max_depth=8)
    multi = mi.render(scene, integrator=mi.load_dict({"type": "path", "max_depth": 8, "samples_per_pass": 4}), spp=16, seed=2).cpu().numpy()
    assert rel_l2(multi, single) > 1e-2                                   # not the same samples as one wavefront of 16 spp
    # lane bands of the per-pass wavefront (what each rank of render_distributed renders)
    integ = mi.load_dict({"type": "path", "max_depth": 8, "samples_per_pass": 4})
    n = 48 * 48 * 4
    whole = integ.render_film(scene, 0, 2, 16)
    film = None
    for lo, hi in ((0, 20 * 48 * 4), (20 * 48 * 4, 20 * 48 * 4 + 501), (20 * 48 * 4 + 501, n)):
        film = integ.render_film(scene, 0, 2, 16, lanes=(lo, hi), film=film)
    torch.cuda.synchronize()
    assert rel_l2(film.cpu().numpy(), whole.cpu().numpy()) < 1e-6
    # max_depth = 0: only the weight channel, jitter = numbers 2p, 2p+1 of each stream
    integ0 = mi.load_dict({"type": "path", "max_depth": 0, "samples_per_pass": 2})
    f0 = integ0.render_film(scene, 0, 7, 6).cpu().numpy()
    r0, _ = osc.render_path_passes(sensor, seed=7, spp=6, spp_per_pass=2, max_depth=0, raw=True)
    assert rel_l2(f0, r0) < 1e-5
    # box filter + crop window
    scene, osc, sensor = cbox(mi, O, 64, crop=(8, 16, 40, 24), rfilter="box")
    integ = mi.load_dict({"type": "path", "max_depth": 5, "samples_per_pass": 2})
    img = mi.render(scene, integrator=integ, spp=8, seed=4).cpu().numpy()
    ref, _ = osc.render_path_passes(sensor, seed=4, spp=8, spp_per_pass=2, max_depth=5)
    assert rel_l2(img, ref) < 1e-4


# ---------------------------------------------------------------- BSDF plugins beyond diffuse

def _material_scene(mi, O, res, integrator=None, smooth=False):
    from tests.test_bsdfs_cpu import _material_cbox, _smooth_material_cbox
    from tests.test_cpu_host import oracle_scene_from
    d = (_smooth_material_cbox if smooth else _material_cbox)(mi, res)
    if integrator:
        d["integrator"] = integrator
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    return scene, osc, sensor


def test_material_scene_forward_parity(mi, O):
    """Cornell box with roughplastic walls, twosided(roughconductor GGX anisotropic), twosided(roughconductor | diffuse) and a
    dielectric (glass) box: `path` and `prb` primal images vs the oracle, north_star forward tolerance"""
    scene, osc, sensor = _material_scene(mi, O, 48)
    img = mi.render(scene, spp=16, seed=2).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=2, spp=16, max_depth=8)
    assert np.isfinite(img).all() and rel_l2(img, ref) < 1e-4
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=16, seed=2).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=2, spp=16, max_depth=6)
    assert rel_l2(img, ref) < 1e-4


def test_material_scene_prb_gradients(mi, O):
    """PRB adjoint w.r.t. the slot-0 colour parameter of every BSDF record (roughplastic.diffuse_reflectance,
    roughconductor.specular_reflectance, diffuse.reflectance of the back side); the dielectric has no gradient (delta lobes)"""
    scene, osc, sensor = _material_scene(mi, O, 40, {"type": "prb", "max_depth": 6})
    integ = scene.integrator()
    grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
    grads = integ.render_backward(scene, None, grad_in, seed=9, spp=16)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=16, max_depth=6)
    keys = {k: v for k, v in scene._param_keys().items() if v[0] != "emit"}
    got = np.stack([grads[k].cpu().numpy() for k in keys]); want = np.stack([g_refl[b.index] for (_, b) in keys.values()])
    assert np.abs(want).max() > 0 and rel_l2(got, want) < 1e-3
    glass = [b for (_, b) in keys.values() if b.kind == "dielectric"]
    assert glass and not grads[[k for k, (_, b) in keys.items() if b is glass[0]][0]].any()


def test_smooth_material_scene_parity_and_gradients(mi, O):
    """Cornell box with `plastic` walls (linear and nonlinear), a twosided gold `conductor` wall and a mirror box: `path`, `prb` primal and the
    PRB adjoint (plastic.diffuse_reflectance; the conductors only have a delta lobe, hence no gradient) vs the oracle"""
    scene, osc, sensor = _material_scene(mi, O, 48, smooth=True)
    img = mi.render(scene, spp=16, seed=2).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=2, spp=16, max_depth=8)
    assert np.isfinite(img).all() and rel_l2(img, ref) < 1e-4
    integ = mi.load_dict({"type": "prb", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=16, seed=2).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=2, spp=16, max_depth=6)
    assert rel_l2(img, ref) < 1e-4
    grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (48, 48, 3)).astype(np.float32)
    grads = integ.render_backward(scene, None, grad_in, seed=9, spp=16)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=16, max_depth=6)
    keys = {k: v for k, v in scene._param_keys().items() if v[0] != "emit"}
    got = np.stack([grads[k].cpu().numpy() for k in keys]); want = np.stack([g_refl[b.index] for (_, b) in keys.values()])
    assert np.abs(want).max() > 0 and rel_l2(got, want) < 1e-3
    for k, (_, b) in keys.items():
        if b.kind == "conductor":
            assert not grads[k].any()


def test_bsdf_plugins_device_vs_oracle(mi, O):
    """array-valued BSDF::eval_pdf / sample of every plugin on the GPU vs the oracle (same inputs)"""
    import ctypes as C
    from tests.test_bsdfs_cpu import BSDF_DICTS
    rng = np.random.default_rng(5); n = 512
    z = rng.uniform(-1, 1, (2, n)); ph = rng.uniform(0, 2 * np.pi, (2, n)); r = np.sqrt(1 - z * z)
    wi = np.stack([r[0] * np.cos(ph[0]), r[0] * np.sin(ph[0]), z[0]]).astype(np.float32)
    wo = np.stack([r[1] * np.cos(ph[1]), r[1] * np.sin(ph[1]), z[1]]).astype(np.float32)
    s1 = rng.random(n).astype(np.float32); s2 = rng.random((2, n)).astype(np.float32)
    types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3, "conductor": 4, "plastic": 5}
    for name, d in BSDF_DICTS.items():
        bsdf = mi.load_dict(d)
        si = type("SI", (), dict(wi=wi, uv=None))()
        val, pdf = bsdf.eval_pdf(mi.BSDFContext(), si, wo)
        bs, w = bsdf.sample(mi.BSDFContext(), si, s1, s2)
        sd = O.SceneData()
        sd.bsdfs = [(types[b.kind], -1, b.value, dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta, eta_c=b.eta_c,
                                                     k_c=b.k_c, back=b.back.index if b.back is not None else -1)) for b in bsdf.scene.bsdf_objs]
        osc = O.OracleScene(sd)
        rv = np.empty((3, n), np.float32); rp = np.empty(n, np.float32); rwo = np.empty((3, n), np.float32); rw = np.empty((3, n), np.float32); rsp = np.empty(n, np.float32)
        uv = O.f32([0, 0])
        for i in range(n):
            v = np.empty(3, np.float32); p = C.c_float(); a = np.ascontiguousarray(wi[:, i]); b_ = np.ascontiguousarray(wo[:, i])
            O.lib().orc_bsdf_eval_pdf(osc.handle, bsdf.index, O.fp(a), O.fp(uv), O.fp(b_), O.fp(v), C.byref(p)); rv[:, i] = v; rp[i] = p.value
            o = np.empty(3, np.float32); ww = np.empty(3, np.float32); eta = C.c_float(); dl = C.c_int(); ss = np.ascontiguousarray(s2[:, i])
            O.lib().orc_bsdf_sample(osc.handle, bsdf.index, O.fp(a), O.fp(uv), C.c_float(float(s1[i])), O.fp(ss), O.fp(o), C.byref(p), O.fp(ww), C.byref(eta), C.byref(dl))
            rwo[:, i] = o; rw[:, i] = ww; rsp[i] = p.value
        # erf / erfinv / exp / log / tan are the product's own polynomial versions (har_math.h) and the oracle restates the same arithmetic (orc_math.h):
        # the device's results are the oracle's BIT FOR BIT, ill-conditioned samples (grazing directions, the Newton iterations of the Beckmann
        # visible-normal sampler) included
        for what, a, b in [("value", val, rv), ("pdf", pdf, rp), ("wo", bs.wo, rwo), ("sample pdf", bs.pdf, rsp), ("weight", w, rw)]:
            a = a.cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), (name, what, int((a != b).sum()), float(np.nanmax(np.abs(a - b))))


def test_constant_environment_emitter_parity(mi, O):
    """`constant` emitter (src/emitters/constant.cpp) + area light in an opened Cornell box, and alone around a cube:
    path / prb primal images and prb gradients vs the oracle"""
    from tests.test_emitters_cpu import env_scene
    from tests.test_cpu_host import oracle_scene_from
    for with_area in (True, False):
        scene = mi.load_dict(env_scene(mi, 40, with_area))
        osc, sensor = oracle_scene_from(O, scene)
        img = mi.render(scene, spp=16, seed=5).cpu().numpy()
        ref, _ = osc.render_path(sensor, seed=5, spp=16, max_depth=8)
        assert rel_l2(img, ref) < 1e-4
        integ = mi.load_dict({"type": "prb", "max_depth": 6})
        img = mi.render(scene, integrator=integ, spp=16, seed=5).cpu().numpy()
        ref, _ = osc.render_prb(sensor, seed=5, spp=16, max_depth=6)
        assert rel_l2(img, ref) < 1e-4
        grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
        grads = integ.render_backward(scene, None, grad_in, seed=3, spp=8)
        g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=3, spp=8, max_depth=6)
        keys = {k: v for k, v in scene._param_keys().items() if v[0] != "emit"}
        got = np.stack([grads[k].cpu().numpy() for k in keys]); want = np.stack([g_refl[b.index] for (_, b) in keys.values()])
        assert rel_l2(got, want) < 1e-3


def test_envmap_emitter_parity(mi, O):
    """`envmap` emitter (src/emitters/envmap.cpp): hierarchical luminance sampling + bilinear lat-long lookup on the device vs the oracle --
    an opened Cornell box (area light + environment) and a cube on a floor lit by the environment alone, rotated map, MIS compensation;
    path / prb primal images, prb gradients, and the materials scene (specular lobes see the map through BSDF sampling)."""
    from tests.test_envmap_cpu import _env_scene
    from tests.test_cpu_host import oracle_scene_from
    T = mi.ScalarTransform4f
    for area, tw, mis in ((True, None, False), (False, T().rotate([0, 1, 0], 70).rotate([0, 0, 1], 20), True)):
        scene, _ = _env_scene(mi, res=48, to_world=tw, mis=mis, with_area_light=area)
        osc, sensor = oracle_scene_from(O, scene)
        img = mi.render(scene, spp=16, seed=5).cpu().numpy()
        ref, st = osc.render_path(sensor, seed=5, spp=16, max_depth=8 if area else 6)
        assert ref.mean() > 0.05 and rel_l2(img, ref) < 1e-4
        gst = scene.integrator().stats()
        assert gst["paths"] == st.paths and gst["vertices"] == st.vertices
        integ = mi.load_dict({"type": "prb", "max_depth": 6})
        img = mi.render(scene, integrator=integ, spp=16, seed=5).cpu().numpy()
        ref, _ = osc.render_prb(sensor, seed=5, spp=16, max_depth=6)
        assert rel_l2(img, ref) < 1e-4
        grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (48, 48, 3)).astype(np.float32)
        grads = integ.render_backward(scene, None, grad_in, seed=3, spp=8)
        g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=3, spp=8, max_depth=6)
        keys = {k: v for k, v in scene._param_keys().items() if v[0] != "emit"}
        got = np.stack([grads[k].cpu().numpy() for k in keys]); want = np.stack([g_refl[b.index] for (_, b) in keys.values()])
        assert rel_l2(got, want) < 1e-3
    # glossy / dielectric materials under an environment map
    from tests.test_bsdfs_cpu import _material_cbox
    d = _material_cbox(mi, 40); d.pop("ceiling", None)
    env = (np.random.default_rng(9).random((16, 32, 3)).astype(np.float32) ** 2); env[4, 20] = [30, 30, 25]
    d["env"] = {"type": "envmap", "bitmap": mi.Bitmap(env), "to_world": T().rotate([0, 1, 0], 200)}
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    img = mi.render(scene, spp=16, seed=1).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=1, spp=16, max_depth=scene.integrator().max_depth)
    assert rel_l2(img, ref) < 1e-4


def test_ztest_product_vs_oracle(mi, O):
    """The reference's statistical render test (src/render/tests/test_renders.py:146-236; tests/ztest.py) with DIFFERENT seeds on both sides:
    complements the same-seed parity tests (it would catch correlated sample streams across chunk / pass / band seams, which same-seed
    comparisons cannot see).  Sample budget 2e6 per image as in the reference."""
    from tests import ztest
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_envmap_cpu import _env_scene
    res = 32; spp = int(2e6) // (res * res)
    spp = 1 << (spp.bit_length() - 1)                   # 1024
    cases = []
    scene, osc, sensor = cbox(mi, O, res, rfilter="box")
    cases.append(("cornell", scene, osc, sensor, 8))
    scene, _ = _env_scene(mi, res=res, to_world=mi.ScalarTransform4f().rotate([0, 1, 0], 70), with_area_light=True)
    d = None
    osc, sensor = oracle_scene_from(O, scene)
    cases.append(("envmap", scene, osc, sensor, 8))
    for name, scene, osc, sensor, md in cases:
        if sensor.rfilter != 0:                           # pixels must be independent: rebuild the sensor with a box filter
            scene.sensors()[0].film().rfilter = 0; scene.sensors()[0].update(); sensor.rfilter = 0
        ref_mean, ref_var, n_ref = ztest.oracle_reference(osc, sensor, spp_b=8, batches=128, max_depth=md)     # many small batches: the variance estimate needs them
        for integ in (mi.load_dict({"type": "path", "max_depth": md}),
                      mi.load_dict({"type": "path", "max_depth": md, "chunk_lanes": 40000}),
                      mi.load_dict({"type": "path", "max_depth": md, "samples_per_pass": spp // 8})):
            img = mi.render(scene, integrator=integ, spp=spp, seed=4242).cpu().numpy()
            ok, pmin, alpha = ztest.accept(img, spp, ref_mean, ref_var, n_ref)
            assert ok, (name, pmin, alpha)
        ok, _, _ = ztest.accept(img * 1.03, spp, ref_mean, ref_var, n_ref)
        assert not ok, name                              # power: +3 % is rejected


def test_mesh_area_emitters_parity(mi, O):
    """area lights on arbitrary triangle meshes (Mesh::sample_position: face pmf + uniform triangle + interpolated normals) vs the oracle:
    path / prb images, BSDF and emitter-radiance gradients"""
    from tests.test_emitters_cpu import mesh_light_scene
    from tests.test_cpu_host import oracle_scene_from
    for normals in (True, False):
        scene = mi.load_dict(mesh_light_scene(mi, 40, normals))
        assert sorted(e["type"] for e in scene.emitters) == [3, 3]
        osc, sensor = oracle_scene_from(O, scene)
        img = mi.render(scene, spp=16, seed=5).cpu().numpy()
        ref, st = osc.render_path(sensor, seed=5, spp=16, max_depth=8)
        assert ref.mean() > 0.02 and rel_l2(img, ref) < 1e-4
        gst = scene.integrator().stats()
        assert gst["paths"] == st.paths and gst["vertices"] == st.vertices      # (the product only traces shadow rays of non-zero contributions)
        integ = mi.load_dict({"type": "prb", "max_depth": 6})
        img = mi.render(scene, integrator=integ, spp=16, seed=5).cpu().numpy()
        ref, _ = osc.render_prb(sensor, seed=5, spp=16, max_depth=6)
        assert rel_l2(img, ref) < 1e-4
        grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
        grads = integ.render_backward(scene, None, grad_in, seed=3, spp=8)
        g_refl, _, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=8, max_depth=6)
        keys = scene._param_keys()
        bk = {k: v for k, v in keys.items() if v[0] != "emit"}; ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
        assert rel_l2(np.stack([grads[k].cpu().numpy() for k in bk]), np.stack([g_refl[b.index] for (_, b) in bk.values()])) < 1e-3
        assert len(ek) == 2 and rel_l2(np.stack([grads[k].cpu().numpy() for k in ek]), np.stack([g_emit[i] for i in ek.values()])) < 1e-3


def test_prb_emitter_radiance_gradients(mi, O):
    """d loss / d radiance of `area` and `constant` emitters (prb.py:160-161 emission term, :198-206 emitter sampling term with the emitter
    attached) vs the oracle, through render_backward and through mi.render + autograd; the BSDF gradients are unchanged by the extra outputs"""
    import torch
    from tests.test_emitters_cpu import env_scene
    from tests.test_cpu_host import oracle_scene_from
    for name, d in (("cornell", None), ("area+constant", env_scene(mi, 40, True)), ("constant", env_scene(mi, 40, False))):
        if d is None:
            d = mi.cornell_box(); d["sensor"]["film"]["width"] = 40; d["sensor"]["film"]["height"] = 40
        scene = mi.load_dict(d)
        osc, sensor = oracle_scene_from(O, scene)
        integ = mi.load_dict({"type": "prb", "max_depth": 6})
        grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
        grads = integ.render_backward(scene, None, grad_in, seed=3, spp=16)
        g_refl, _, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=16, max_depth=6)
        keys = scene._param_keys()
        ek = {k: v[1] for k, v in keys.items() if v[0] == "emit"}
        assert len(ek) == len(scene.emitters) >= 1, name
        got = np.stack([grads[k].cpu().numpy() for k in ek]); want = np.stack([g_emit[i] for i in ek.values()])
        assert np.abs(want).min() > 0 and rel_l2(got, want) < 1e-3, name
        bk = {k: v for k, v in keys.items() if v[0] != "emit"}
        got = np.stack([grads[k].cpu().numpy() for k in bk]); want = np.stack([g_refl[b.index] for (_, b) in bk.values()])
        assert rel_l2(got, want) < 1e-3, name
        # switching the emitter outputs off leaves the BSDF gradients as they were
        integ2 = mi.load_dict({"type": "prb", "max_depth": 6, "emitter_gradients": False})
        grads2 = integ2.render_backward(scene, None, grad_in, seed=3, spp=16)
        assert not any(k in grads2 for k in ek)
        assert rel_l2(np.stack([grads2[k].cpu().numpy() for k in bk]), got) < 1e-5
    # autograd: the loss is linear in the radiance, so radiance . grad == loss of the emitted part; check against a finite difference
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    d["integrator"] = {"type": "prb", "max_depth": 6}
    scene = mi.load_dict(d)
    params = mi.traverse(scene); key = "light.emitter.radiance.value"
    assert key in params
    def loss_of(rad):
        params[key] = torch.tensor(rad, dtype=torch.float32, device="cuda"); params.update()
        return float((mi.render(scene, spp=64, seed=0) ** 2).mean())
    base = params[key].detach().cpu().numpy().copy()
    p = params[key].detach().clone().requires_grad_(); params[key] = p
    img = mi.render(scene, params, spp=64, seed=0, seed_grad=11)
    (img ** 2).mean().backward()
    g = p.grad.cpu().numpy()
    for c in range(3):
        e = np.zeros(3, np.float32); e[c] = 0.05 * base[c]
        fd = (loss_of(base + e) - loss_of(base - e)) / (2 * e[c])
        assert abs(g[c] - fd) < 0.03 * abs(fd) + 1e-6, (c, g[c], fd)      # different sample sets in the primal / adjoint passes: statistical agreement


def test_prb_replay_cache_is_transparent(mi, O):
    """the adjoint pass with the replay cache (primal hit / visibility records reused) gives the gradients of a full re-trace,
    for constant colours, textures and the all-materials scene, also when the path is deeper than the cache"""
    scenes = [lambda: mi.textured_cornell_box(res=40, tex_res=16, spp=8), lambda: __import__("tests.test_bsdfs_cpu", fromlist=["x"])._material_cbox(mi, 40)]
    for mk in scenes:
        for md in (6, 20):
            got = []
            for cache in (True, False):
                d = mk(); d["integrator"] = {"type": "prb", "max_depth": md, "rr_depth": 5, "replay_cache": cache}
                scene = mi.load_dict(d)
                grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
                grads = scene.integrator().render_backward(scene, None, grad_in, seed=2, spp=8)
                got.append(np.concatenate([g.cpu().numpy().ravel() for g in grads.values()]))
            assert np.abs(got[1]).max() > 0 and rel_l2(got[0], got[1]) < 1e-5


# ---------------------------------------------------------------- vertex-position gradients

def _grid_floor(mi, d, n):
    """replace the Cornell floor by an n x n vertex grid (flat, no normals): more differentiated vertices than the LDS accumulator holds"""
    floor = mi.load_dict({"type": "rectangle", "to_world": mi.cornell_box()["floor"]["to_world"]})
    P0 = floor.V[0, :3]
    # corners of the rectangle lowering: find two edge vectors from vertex 0
    e = [floor.V[k, :3] - P0 for k in range(1, 4)]
    e.sort(key=lambda v: np.linalg.norm(v)); ex, ey = e[0], e[1]
    nrm = np.cross(floor.V[floor.F[0, 1], :3] - floor.V[floor.F[0, 0], :3], floor.V[floor.F[0, 2], :3] - floor.V[floor.F[0, 0], :3])
    if np.dot(np.cross(ex, ey), nrm) < 0:
        ex, ey = ey, ex
    s = np.linspace(0, 1, n, dtype=np.float32)
    P = np.stack([P0 + a * ex + b * ey for b in s for a in s]).astype(np.float32)
    F = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i; F += [[a, a + 1, a + n + 1], [a, a + n + 1, a + n]]
    d["floor"] = {"type": "mesh", "positions": P, "faces": np.asarray(F, np.uint32), "bsdf": {"type": "ref", "id": "white"}}
    return d


@pytest.mark.parametrize("which", ["slab", "slab_textured", "slab_env", "slab_twosided", "slab_crop_box", "cbox", "cbox_grid", "cbox_nocache", "slab_rough_conductor", "slab_rough_plastic",
                                   "floor_roughconductor", "floor_roughconductor_beckmann", "floor_roughplastic", "floor_plastic", "both_roughconductor", "floor_roughconductor_aniso", "cbox_rough",
                                   "smooth_floor", "smooth_floor_roughplastic", "cbox_shapes", "smooth_spheres"])
def test_prb_vertex_position_gradients(mi, O, which):
    """har_integrator_set_grad_positions: the wavefront adjoint (k_shade<ADJOINT, SHAPE> geometry records, visibility from k_resolve,
    k_shape_adjoint with the next bounce's detached interaction) vs the oracle's dual-number restatement, vertex by vertex; the colour
    gradients of the same call must not change"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_shape_gradients_cpu import slab_scene, cbox_mesh_scene, twosided_slab_scene, rough_slab_scene, smooth_slab_scene, mesh_index
    if which.startswith("smooth_floor"):      # vertex normals regenerated from the positions (mesh.cpp:876-878): k_shape_adjoint's normal adjoints + k_normals_adjoint
        res = 24; d = smooth_slab_scene(mi, res, model=which[13:] or None); names = ["floor", "ceiling"]
    elif which == "smooth_spheres":           # closed smooth meshes (UV spheres with seam and pole vertices) inside the Cornell box; normals regenerated below
        res = 32; d = mi.instanced_spheres_scene(width=res, height=res, spp=16, grid=2, n_u=16, n_v=8, flatten=True); names = ["ball000", "ball001", "ball003", "floor"]      # (ball002 is out of every path's reach at this size)
    elif which == "cbox_shapes":              # the Cornell box's own rectangles and cubes: meshes WITH vertex normals (equal to the regenerated ones on flat faces)
        res = 32; d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res; names = ["small-box", "large-box", "floor", "back"]
    elif which == "slab_twosided":
        res = 24; d = twosided_slab_scene(mi, res); names = ["floor", "ceiling", "sheet"]
    elif which.startswith("slab_rough"):      # a rough (not differentiated) ceiling: its vertices follow the floor through the attached si.wi (prb.py:128-140)
        res = 24; d = rough_slab_scene(mi, res, "roughconductor" if which.endswith("conductor") else "roughplastic"); names = ["floor"]
    elif which.startswith("floor_") or which.startswith("both_"):      # the MOVING meshes carry the non-diffuse models: d f / d wi, d f / d wo (har_bsdf_dir.h)
        where, model = which.split("_", 1)
        res = 24; d = rough_slab_scene(mi, res, model, where); names = ["floor", "ceiling"]
    elif which.startswith("cbox"):
        res = 32; d = cbox_mesh_scene(mi, res); names = ["small-box", "large-box", "floor"]
        if which == "cbox_grid":
            d = _grid_floor(mi, d, 36)
        if which == "cbox_rough":              # every model on moving geometry at once: a textured rough-plastic floor, a twosided conductor box, a plastic box
            from tests.test_shape_gradients_cpu import ROUGH_BSDFS
            d["floor"]["bsdf"] = {"type": "roughplastic", "alpha": 0.25, "diffuse_reflectance": d["floor"]["bsdf"]["reflectance"]}
            d["small-box"]["bsdf"] = {"type": "twosided", "bsdf": dict(ROUGH_BSDFS["roughconductor"])}
            d["large-box"]["bsdf"] = dict(ROUGH_BSDFS["plastic"])
    else:
        res = 24; d = slab_scene(mi, res, textured=which in ("slab_textured", "slab_crop_box"), env=which == "slab_env"); names = ["floor"] + ([] if which == "slab_env" else ["ceiling"])
    spp = 16
    if which == "slab_crop_box":          # ragged case: crop window, box filter, sample count that is not a power of two
        d["sensor"]["film"].update({"width": 40, "height": 30, "crop_offset_x": 9, "crop_offset_y": 4, "crop_width": res, "crop_height": res, "rfilter": {"type": "box"}})
        spp = 12
    d["integrator"] = {"type": "prb", "max_depth": 5, "shape_gradients": [n + ".positions" for n in names]}
    if which == "cbox_nocache":
        d["integrator"]["replay_cache"] = False
    scene = mi.load_dict(d)
    if which == "smooth_spheres":             # writing the positions regenerates the vertex normals (mesh.cpp:876-878): the analytic normals of the scene give way
        params = mi.traverse(scene)
        for n in names[:3]:
            params[n + ".positions"] = params[n + ".positions"].clone()
        params.update()
    osc, sensor = oracle_scene_from(O, scene)
    ids = [mesh_index(scene, n) for n in names]
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    integ = scene.integrator()
    grads = integ.render_backward(scene, None, grad_in, seed=3, spp=spp)
    want, w_refl, w_tex, _ = osc.render_prb_backward_shape(sensor, grad_in, ids, seed=3, spp=spp, max_depth=5)
    for n, m in zip(names, ids):
        got = grads[n + ".positions"].cpu().numpy().reshape(-1, 3)
        scale = np.abs(want[m]).max()
        assert scale > 0 and np.abs(got - want[m]).max() < 2e-3 * scale, (which, n, np.abs(got - want[m]).max() / scale)
    keys = {k: v for k, v in scene._param_keys().items() if v[0] != "emit"}
    for k, (kind, b) in keys.items():
        ref = w_tex[b.tex_index] if kind == "tex" else w_refl[b.index]
        if not ref.any():
            assert not grads[k].any(), k                # e.g. the black BSDF of the light
            continue
        assert rel_l2(grads[k].cpu().numpy(), ref) < 1e-3, k
    # switching the feature off again restores the plain adjoint
    integ.shape_gradients = False
    plain = integ.render_backward(scene, None, grad_in, seed=3, spp=spp)
    assert not any(k.endswith(".positions") for k in plain)
    for k in keys:
        assert np.allclose(plain[k].cpu().numpy(), grads[k].cpu().numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("which", ["point", "spot", "directional", "spot_rough", "point_and_area", "meshlight", "texlight", "envmap", "weighted"])
def test_prb_vertex_position_gradients_generic_emitters(mi, O, which):
    """vertex-position gradients in scenes of the generic-emitter kernel class (round 6): point / spot lights (prb.py:191-192 re-attaches ds.d = normalize(ds.p - si.p);
    the point light's 1 / r^2 follows si.p, point.cpp:155-165; the spot's falloff follows ds.d, its rcp(ds.dist) is detached, spot.cpp:252-274), directional lights and
    environment maps (EmitterFlags::Infinite: nothing re-attached), triangle-mesh and bitmap-radiance area lights (surface: Jacobian + direction), weighted emitters --
    k_shade<ADJOINT, ALL | ENVMAP (| TEXLIGHT), SHAPE> + k_shape_adjoint against the oracle's dual numbers, vertex by vertex"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_shape_gradients_cpu import slab_scene, delta_slab_scene, generic_light_slab_scene, mesh_index
    res = 24
    if which in ("meshlight", "texlight", "envmap", "weighted"):
        d = generic_light_slab_scene(mi, res, which)
    else:
        d = delta_slab_scene(mi, res, which.split("_")[0], which.endswith("_rough"))
    if which == "point_and_area":
        d["light"] = slab_scene(mi, res)["light"]
    names = ["floor"] + (["ceiling"] if "ceiling" in d else [])
    d["integrator"] = {"type": "prb", "max_depth": 5, "shape_gradients": [n + ".positions" for n in names]}
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    ids = [mesh_index(scene, n) for n in names]
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    integ = scene.integrator()
    grads = integ.render_backward(scene, None, grad_in, seed=3, spp=16)
    want, w_refl, w_tex, _ = osc.render_prb_backward_shape(sensor, grad_in, ids, seed=3, spp=16, max_depth=5)
    for n, m in zip(names, ids):
        got = grads[n + ".positions"].cpu().numpy().reshape(-1, 3)
        scale = np.abs(want[m]).max()
        assert scale > 0 and np.abs(got - want[m]).max() < 2e-3 * scale, (which, n, np.abs(got - want[m]).max() / scale)
    for k, (kind, b) in scene._param_keys().items():
        if kind not in ("tex", "refl"):
            continue
        ref = w_tex[b.tex_index] if kind == "tex" else w_refl[b.index]
        if not ref.any():
            assert not grads[k].any(), k
            continue
        assert rel_l2(grads[k].cpu().numpy(), ref) < 1e-3, k


def test_prb_instance_gradients_under_a_point_light(mi, O):
    """instance to_world gradients (instance.cpp:150-266) with the point light's re-attached direction and 1 / r^2"""
    from tests.test_shape_gradients_cpu import instanced_slab_scene
    res = 24
    d = instanced_slab_scene(mi, res)
    d.pop("light")
    d["lamp"] = {"type": "point", "position": [0.1, 2.5, 0.2], "intensity": {"type": "rgb", "value": [42.0, 39.0, 34.0]}}
    keys = [k for k, v in d.items() if isinstance(v, dict) and v.get("type") == "instance"]
    d["integrator"] = {"type": "prb", "max_depth": 5, "shape_gradients": [k + ".to_world" for k in keys]}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=16)
    want, _, _, _ = osc.render_prb_backward_instances(sensor, grad_in, None, seed=3, spp=16, max_depth=5)
    for i, k in enumerate(keys):
        got = grads[k + ".to_world"].cpu().numpy()
        scale = np.abs(want[i]).max()
        assert scale > 0 and np.abs(got[:3] - want[i]).max() < 2e-3 * scale, (k, np.abs(got[:3] - want[i]).max() / scale)


@pytest.mark.parametrize("which", ["black_walls", "black_texture", "dark_light", "black_walls_nocache"])
def test_prb_gradients_at_zero_parameters(mi, O, which):
    """a parameter that is exactly ZERO still has a derivative: d Lr_dir / d rho = beta mis (cos / pi) em_weight at rho = 0, d Lr_dir / d radiance at radiance = 0.  The reference tests
    the visibility of every emitter sample with a density (scene.cpp:338-346), so the terms exist there; until round 6 the wavefront filed no shadow ray for a sample whose
    CONTRIBUTION was zero and lost them (an albedo texture initialised to black would never have left black).  Record tape and replay cache, against the oracle"""
    res = 24
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["integrator"] = {"type": "prb", "max_depth": 4}
    if which.startswith("black_walls"):
        d["white"]["reflectance"]["value"] = [0.0, 0.0, 0.0]
    elif which == "black_texture":
        t = np.random.default_rng(3).uniform(0.2, 0.8, (6, 6, 3)).astype(np.float32); t[:4, :4] = 0.0
        d["white"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": t, "raw": True, "filter_type": "nearest"}}
    else:          # a second light that is switched off: its radiance gradient says what switching it on would do
        T = mi.ScalarTransform4f
        d["lamp2"] = {"type": "rectangle", "to_world": T().translate([0.4, -0.3, 0.2]).rotate([0, 1, 0], -70.0).scale([0.15, 0.2, 1.0]),
                      "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.3, 0.3, 0.3]}},
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [0.0, 0.0, 0.0]}}}
    if which.endswith("nocache"):
        d["integrator"]["replay_cache"] = False
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=32)
    w_refl, w_tex, w_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=32, max_depth=4)
    checked = 0
    for k, (kind, b) in scene._param_keys().items():
        ref = w_emit[b] if kind == "emit" else (w_tex[b.tex_index] if kind == "tex" else w_refl[b.index])
        if np.any(ref):
            assert rel_l2(grads[k].cpu().numpy(), ref) < 1e-3, (which, k, grads[k].cpu().numpy(), ref)
            checked += 1
    key = {"black_walls": "white.reflectance.value", "black_walls_nocache": "white.reflectance.value", "black_texture": "white.reflectance.data", "dark_light": "lamp2.emitter.radiance.value"}[which]
    g = grads[key].cpu().numpy()
    assert checked >= 2 and np.abs(g).max() > 0
    if which == "black_texture":
        assert np.abs(g[:3, :3]).max() > 0


def test_vertex_position_update_rebuilds_the_scene(mi, O):
    """params['floor.positions'] = ...; params.update(): the next render sees the moved mesh (and matches the oracle's)"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_shape_gradients_cpu import slab_scene, mesh_index
    scene = mi.load_dict(slab_scene(mi, 24))
    params = mi.traverse(scene)
    key = "floor.positions"
    assert key in params and params[key].numel() == 12
    before = mi.render(scene, spp=8, seed=1).cpu().numpy()
    p = params[key].clone().reshape(-1, 3); p[:, 1] += 0.4
    params[key] = p.reshape(-1); params.update()
    after = mi.render(scene, spp=8, seed=1).cpu().numpy()
    osc, sensor = oracle_scene_from(O, scene)          # built from the updated arrays
    ref, _ = osc.render_prb(sensor, seed=1, spp=8, max_depth=4)
    assert rel_l2(after, ref) < 1e-4 and rel_l2(after, before) > 1e-2


def test_vertex_position_gradients_refused_outside_their_domain(mi):
    """purely specular BSDFs on moving geometry / meshes with vertex normals: an error, not a silently incomplete gradient"""
    from tests.test_bsdfs_cpu import _material_cbox
    g = np.ones((16, 16, 3), np.float32)
    # a mesh whose vertex normals are NOT the ones a position update regenerates (analytic normals of the bumpy sphere): refused when named, left out by `True`,
    # accepted once its positions have been written (params.update() regenerates the normals, mesh.cpp:876-878)
    g = np.ones((32, 32, 3), np.float32)
    d = mi.instanced_spheres_scene(width=32, height=32, spp=4, grid=2, n_u=12, n_v=6, flatten=True)
    d["integrator"] = {"type": "prb", "max_depth": 3, "shape_gradients": ["ball000.positions"]}
    scene = mi.load_dict(d)
    with pytest.raises(RuntimeError, match="regenerates"):
        scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    scene.integrator().shape_gradients = True
    out = scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    assert "ball000.positions" not in out and "floor.positions" in out          # the box's rectangles: flat faces, regenerated == stored
    params = mi.traverse(scene)
    for k in range(4):
        params["ball%03d.positions" % k] = params["ball%03d.positions" % k].clone()
    params.update()
    out = scene.integrator().render_backward(scene, None, g, seed=0, spp=16)
    assert all("ball%03d.positions" % k in out for k in range(4)) and max(float(out["ball%03d.positions" % k].abs().max()) for k in range(4)) > 0
    with pytest.raises(KeyError):
        scene.integrator().shape_gradients = ["nonexistent.positions"]
        scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    from tests.test_shape_gradients_cpu import slab_scene
    g = np.ones((16, 16, 3), np.float32)
    # a mesh with only delta lobes may be PART of the scene; asking for ITS vertex positions is refused (eval() is zero: prb.py:288 would form relative_grad(0)),
    # `True` selects the meshes the adjoint can differentiate -- rough models included
    d = slab_scene(mi, 16); d["ceiling"]["bsdf"] = {"type": "conductor", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]}
    d["integrator"] = {"type": "prb", "max_depth": 3, "shape_gradients": ["ceiling.positions"]}
    scene = mi.load_dict(d)
    with pytest.raises(RuntimeError, match="delta lobes"):
        scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    scene.integrator().shape_gradients = True
    out = scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    assert "floor.positions" in out and "ceiling.positions" not in out
    d["ceiling"]["bsdf"] = {"type": "roughconductor", "alpha": 0.2}
    scene = mi.load_dict(d); scene.integrator().shape_gradients = True
    out = scene.integrator().render_backward(scene, None, g, seed=0, spp=4)
    assert "floor.positions" in out and "ceiling.positions" in out


def test_hide_emitters_parity(mi, O):
    """Integrator property `hide_emitters`: the device round trip of Integrator::skip_area_emitters (continuation lists, re-trace, hit replacement)
    and the hidden environment, `path` / `prb` images and prb gradients vs the oracle; a stack of three emitters along the camera rays"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_emitters_cpu import hide_emitters_scene
    T = mi.ScalarTransform4f
    for stacked in (False, True):
        d = hide_emitters_scene(mi, 40)
        if stacked:       # two more (small, dim) area lights in front of the camera: camera rays cross up to three emitters in a row
            for k, z in enumerate((2.0, 1.2)):
                d["panel%d" % k] = {"type": "rectangle", "to_world": T().translate([0.1 * k, 0.2, z]).scale([0.5, 0.4, 1.0]),
                                    "bsdf": {"type": "ref", "id": "white"}, "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [0.5, 0.4 + k, 0.3]}}}
        for itype, md in (("path", 6), ("prb", 5)):
            d["integrator"] = {"type": itype, "max_depth": md, "hide_emitters": True}
            scene = mi.load_dict(d)
            osc, sensor = oracle_scene_from(O, scene); osc.set_hide_emitters(True)
            img = mi.render(scene, spp=16, seed=5).cpu().numpy()
            ref, _ = (osc.render_path if itype == "path" else osc.render_prb)(sensor, seed=5, spp=16, max_depth=md)
            assert np.isfinite(img).all() and rel_l2(img, ref) < 1e-4, (stacked, itype)
            if itype == "prb":
                grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
                grads = scene.integrator().render_backward(scene, None, grad_in, seed=9, spp=16)
                g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=16, max_depth=md)
                keys = {k: v for k, v in scene._param_keys().items() if v[0] == "rgb"}
                got = np.stack([grads[k].cpu().numpy() for k in keys]); want = np.stack([g_refl[b.index] for (_, b) in keys.values()])
                assert rel_l2(got, want) < 1e-3, stacked
        shown = mi.load_dict({**d, "integrator": {"type": "path", "max_depth": 6}})
        assert rel_l2(mi.render(shown, spp=16, seed=5).cpu().numpy(), img) > 0.1           # the property does change the picture


@pytest.mark.parametrize("name", ["tent_wide", "mitchell_bc", "catmullrom", "lanczos", "lanczos2"])
def test_reconstruction_filters_parity(mi, O, name):
    """tent / mitchell / catmullrom / lanczos film filters (src/rfilters/*.cpp): the splat gather and the adjoint's footprint gather use the same
    per-axis weights as the oracle, negative lobes included; `path` image, `prb` image and texture gradient"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_rfilters_cpu import FILTERS
    d = mi.textured_cornell_box(res=40, tex_res=8, spp=8)
    d["sensor"]["film"]["rfilter"] = FILTERS[name][4]
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    img = mi.render(scene, integrator=mi.load_dict({"type": "path", "max_depth": 6}), spp=8, seed=2).cpu().numpy()
    ref, _ = osc.render_path(sensor, seed=2, spp=8, max_depth=6)
    assert np.isfinite(img).all() and rel_l2(img, ref) < 1e-4
    integ = mi.load_dict({"type": "prb", "max_depth": 5})
    grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
    grads = integ.render_backward(scene, None, grad_in, seed=9, spp=8)
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=8, max_depth=5)
    assert rel_l2(grads["white.reflectance.data"].cpu().numpy(), g_tex[0]) < 1e-3


@pytest.mark.parametrize("rf,crop,spp", [("gaussian", None, 8), ("gaussian", (3, 6, 30, 21), 8), ("tent", None, 4), ("gaussian", None, 256), ("box", None, 8)],
                         ids=["gaussian", "gaussian-crop", "tent", "gaussian-256spp", "box"])
def test_sample_border_parity(mi, O, rf, crop, spp):
    """Film::sample_border (film.cpp:29-32): render()'s lane map covers crop_size + 2 * rfilter->border_size() pixels, shifted back by the border
    (integrator.cpp:162-165, 322-339), splats are clipped to the film.  `path` image (both splat paths: the gather of k_splat at 256 spp holds ONE
    pixel per block, below that several), `prb` image, texture gradients and the sample count vs the oracle; distributed bands are whole rows of the
    SAMPLE grid (band union == whole)."""
    from tests.test_cpu_host import oracle_scene_from
    res = 36
    d = mi.textured_cornell_box(res=res, tex_res=8, spp=spp)
    f = d["sensor"]["film"]; f["rfilter"] = {"type": rf}; f["sample_border"] = True
    if crop:
        f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = crop
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    w, h = (crop[2], crop[3]) if crop else (res, res)
    b = {"gaussian": 2, "tent": 1, "box": 0}[rf]
    integ = mi.load_dict({"type": "path", "max_depth": 6})
    img = mi.render(scene, integrator=integ, spp=spp, seed=2).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=2, spp=spp, max_depth=6)
    assert integ.stats()["paths"] == st.paths == (w + 2 * b) * (h + 2 * b) * spp
    assert img.shape == ref.shape == (h, w, 3) and np.isfinite(img).all() and rel_l2(img, ref) < 1e-4
    # two lane bands of whole sample-grid rows add up to the whole film
    gw, gh = scene.sensors()[0].film().sample_grid()
    cut = (gh // 2) * gw * spp
    parts = integ.render_film(scene, scene.sensors()[0], 2, spp, lanes=(0, cut)) + integ.render_film(scene, scene.sensors()[0], 2, spp, lanes=(cut, gw * gh * spp))
    whole = integ.render_film(scene, scene.sensors()[0], 2, spp)
    assert rel_l2(parts.cpu().numpy(), whole.cpu().numpy()) < 1e-6
    if spp > 16:
        return
    prb = mi.load_dict({"type": "prb", "max_depth": 5})
    img_p = mi.render(scene, integrator=prb, spp=spp, seed=4).cpu().numpy()
    ref_p, _ = osc.render_prb(sensor, seed=4, spp=spp, max_depth=5)
    assert rel_l2(img_p, ref_p) < 1e-4
    grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (h, w, 3)).astype(np.float32)
    grads = prb.render_backward(scene, None, grad_in, seed=9, spp=spp)
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=spp, max_depth=5)
    assert rel_l2(grads["white.reflectance.data"].cpu().numpy(), g_tex[0]) < 1e-3


def test_vertex_position_optimisation_converges(mi):
    """end to end: a floor displaced by 0.35 is pulled back to the height that produced the target image by gradient descent on
    '<mesh>.positions' (render -> d loss / d image -> render_backward -> params.update(), which rebuilds the acceleration structure)"""
    import torch
    from tests.test_shape_gradients_cpu import slab_scene
    res, spp = 32, 32
    d = slab_scene(mi, res); d["integrator"] = {"type": "prb", "max_depth": 4, "shape_gradients": ["floor.positions"], "emitter_gradients": False}
    scene = mi.load_dict(d); integ = scene.integrator()
    target = mi.render(scene, integrator=integ, spp=256, seed=100)
    params = mi.traverse(scene); key = "floor.positions"
    base = params[key].clone().reshape(-1, 3)
    height = torch.tensor(0.35, device=base.device)
    losses, heights = [], []
    for it in range(30):
        p = base.clone(); p[:, 1] += height
        params[key] = p.reshape(-1); params.update()
        img = mi.render(scene, integrator=integ, spp=spp, seed=it)
        diff = img - target
        losses.append(float((diff ** 2).mean())); heights.append(float(height))
        grads = integ.render_backward(scene, None, (2.0 * diff / diff.numel()).cpu().numpy(), seed=it, spp=spp)
        g_h = grads[key].reshape(-1, 3)[:, 1].sum()                    # chain rule: every vertex moves with the height
        height = height - torch.sign(g_h) * 0.06 * 0.9 ** it            # signed steps of decaying length: only the gradient's sign is trusted
    assert abs(heights[-1]) < 0.03 and losses[-1] < 0.1 * losses[0], (heights, losses)


@pytest.mark.parametrize("fmt", ["luminance", "luminance_alpha", "xyz", "xyza"])
def test_film_pixel_formats(mi, O, fmt):
    """HDRFilm pixel formats (hdrfilm.cpp:149-176, develop :326-395): Y = luminance(rgb), XYZ = srgb_to_xyz(rgb) (spectrum.h:402-442) of the ORACLE's image, the
    alpha plane as for rgba; the `prb` adjoint takes a gradient in the film's own channels (the transform is linear: its transpose reaches the RGB adjoint)"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_emitters_cpu import hide_emitters_scene
    d = hide_emitters_scene(mi, 32)
    d["sensor"]["film"]["pixel_format"] = fmt
    d["integrator"] = {"type": "prb", "max_depth": 4}
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    img = mi.render(scene, spp=16, seed=3).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=3, spp=16, max_depth=4)
    y = np.float32(0.212671) * ref[..., 0] + np.float32(0.715160) * ref[..., 1] + np.float32(0.072169) * ref[..., 2]
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
    want = y[..., None] if fmt.startswith("luminance") else ref @ M.T
    nc = want.shape[2]
    assert img.shape == (32, 32, nc + (1 if fmt.endswith("a") else 0))
    assert rel_l2(img[..., :nc], want) < 1e-4
    if fmt.endswith("a"):
        osc.set_alpha_only(True); alpha, _ = osc.render_prb(sensor, seed=3, spp=16, max_depth=4); osc.set_alpha_only(False)
        assert np.abs(img[..., nc] - alpha[..., 0]).max() < 1e-5
    # gradients: sum(w * colour) = sum((M^T w) * rgb)
    w = np.random.default_rng(1).uniform(0.5, 1.5, img.shape).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, w, seed=5, spp=8)
    w_rgb = (w[..., :nc] * np.array([0.212671, 0.715160, 0.072169], np.float32)) if nc == 1 else w[..., :nc] @ M
    g_refl, _, _ = osc.render_prb_backward(sensor, np.ascontiguousarray(w_rgb, np.float32), seed=5, spp=8, max_depth=4)
    keys = [k for k in grads if k.endswith("reflectance.value")]
    assert keys
    for k in keys:
        b = scene._param_keys()[k][1]
        if g_refl[b.index].any():
            assert rel_l2(grads[k].cpu().numpy(), g_refl[b.index]) < 1e-3, k


@pytest.mark.parametrize("config", ["path", "prb", "path_hidden", "path_chunks", "path_passes"])
def test_rgba_film_alpha_parity(mi, O, config):
    """pixel_format = rgba (har_integrator_set_alpha_film): RGB as before, A = filtered valid-sample mask, vs the oracle"""
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_emitters_cpu import hide_emitters_scene
    d = hide_emitters_scene(mi, 40)
    d["sensor"]["film"]["pixel_format"] = "rgba"
    itype = "prb" if config == "prb" else "path"
    d["integrator"] = {"type": itype, "max_depth": 5}
    if config == "path_hidden":
        d["integrator"]["hide_emitters"] = True
    if config == "path_chunks":
        d["integrator"]["chunk_lanes"] = 4096
    if config == "path_passes":
        d["integrator"]["samples_per_pass"] = 4
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene); osc.set_hide_emitters(config == "path_hidden")
    img = mi.render(scene, spp=16, seed=3).cpu().numpy()
    assert img.shape == (40, 40, 4)
    if config == "path_passes":
        ref, _ = osc.render_path_passes(sensor, seed=3, spp=16, spp_per_pass=4, max_depth=5)
        osc.set_alpha_only(True); alpha, _ = osc.render_path_passes(sensor, seed=3, spp=16, spp_per_pass=4, max_depth=5)
    else:
        fn = osc.render_prb if itype == "prb" else osc.render_path
        ref, _ = fn(sensor, seed=3, spp=16, max_depth=5)
        osc.set_alpha_only(True); alpha, _ = fn(sensor, seed=3, spp=16, max_depth=5)
    osc.set_alpha_only(False)
    assert rel_l2(img[..., :3], ref) < 1e-4
    assert np.abs(img[..., 3] - alpha[..., 0]).max() < 1e-5, config
    if config in ("prb", "path_hidden"):
        assert img[..., 3].min() < 0.05            # the sky is a hole
    # the adjoint accepts an rgba-shaped gradient and ignores its alpha plane
    if itype == "prb":
        g4 = np.random.default_rng(1).uniform(0.5, 1.5, (40, 40, 4)).astype(np.float32)
        a = scene.integrator().render_backward(scene, None, g4, seed=9, spp=8)
        b = scene.integrator().render_backward(scene, None, g4[..., :3].copy(), seed=9, spp=8)
        for k in a:
            assert np.allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=1e-5, atol=1e-8)


ROUGH_MODELS = {
    "roughplastic_beckmann": {"type": "roughplastic", "distribution": "beckmann", "alpha": 0.2},
    "roughplastic_ggx": {"type": "roughplastic", "distribution": "ggx", "alpha": 0.2},
    "roughconductor_beckmann": {"type": "roughconductor", "distribution": "beckmann", "alpha": 0.15, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
    "roughconductor_beckmann_aniso": {"type": "roughconductor", "distribution": "beckmann", "alpha_u": 0.05, "alpha_v": 0.3, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
    "roughconductor_ggx_aniso": {"type": "roughconductor", "distribution": "ggx", "alpha_u": 0.05, "alpha_v": 0.3, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
    "roughconductor_beckmann_all_normals": {"type": "roughconductor", "distribution": "beckmann", "alpha": 0.15, "sample_visible": False, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
    "plastic": {"type": "plastic"},
}


@pytest.mark.parametrize("model", sorted(ROUGH_MODELS))
def test_paths_through_rough_bsdfs_are_the_oracles_paths(mi, O, model):
    """Cornell box whose walls and boxes carry one rough / layered model: `path` takes the same discrete decisions (lobe choice, Russian roulette,
    which triangle a grazing ray meets) on the device as in the oracle -- equal vertex counts, image to float-summation order.  This holds because
    no elementary function on the path comes from a maths library (tools/rough_identity.py prints the counts)."""
    d = mi.cornell_box()
    d["white"] = dict(ROUGH_MODELS[model])
    d["sensor"]["film"]["width"] = d["sensor"]["film"]["height"] = 96
    d["sensor"]["sampler"] = {"type": "independent", "sample_count": 16}
    d["integrator"] = {"type": "path", "max_depth": 8, "rr_depth": 5}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=16, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=16, max_depth=8)
    gst = scene.integrator().stats()
    assert gst["paths"] == st.paths and gst["vertices"] == st.vertices, (gst, st.vertices)
    assert rel_l2(img, ref) < 2e-6


def test_golden_textured_area_light(mi):
    """a rectangle light with a bitmap radiance against the committed oracle arrays (no oracle call at run time): forward image and prb image"""
    from tests.test_textured_area_light_cpu import lit_box, _bitmap
    fx = _fx()
    d = lit_box(mi, _bitmap(5), 24, wrap_mode="mirror")
    d["integrator"] = {"type": "path", "max_depth": 6}
    img = mi.render(mi.load_dict(d), spp=16, seed=3).cpu().numpy()
    assert rel_l2(img, fx["texlight_path"]) < 1e-4
    d["integrator"] = {"type": "prb", "max_depth": 5}
    img = mi.render(mi.load_dict(d), spp=16, seed=3).cpu().numpy()
    assert rel_l2(img, fx["texlight_prb"]) < 1e-4
