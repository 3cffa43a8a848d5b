"""GPU tests of what the array-valued entry points take BESIDES their arrays: the reference's `Mask active`, the `BSDFContext` of the BSDF calls
and the `ray_flags` of Scene::ray_intersect / compute_surface_interaction (src/render/scene.cpp:197-244, include/mitsuba/render/bsdf.h:140-186,322-465,
include/mitsuba/render/interaction.h:19-87,804-829).  Everything goes Python mirror -> ctypes -> C ABI (include/hip_ad_rgb.h) -> HIP kernels and is compared with
the oracle's restatement of the same arguments (oracle/orc_bsdf_ctx.h, orc_surface_interaction_flags, orc_*_masked)."""
import json
import os

import numpy as np
import pytest

from tests.test_gpu_boundary import _rays, _scenes, rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ALL, NONE = 0x1ff, 0xffffffff


def test_masked_ray_queries_vs_oracle(mi, O):
    """masked lanes are not traced: t = inf with zero-initialised indices / ray_test false; the other lanes are untouched by their neighbours' masks"""
    n = 50000
    for name, d in _scenes(mi, O):
        scene = mi.load_dict(d)
        osc, _ = O.scene_from_product(scene)
        o, dd, maxt = _rays(n, 21)
        active = np.random.default_rng(8).random(n) < 0.6
        ray = mi.Ray3f(o, dd, maxt)
        want = osc.ray_intersect_masked(o, dd, maxt, active, naive=True)
        for naive in (False, True):
            pi = scene._intersect(ray, naive, active)
            got = [pi.t, pi.prim_uv[0], pi.prim_uv[1], pi.prim_index, pi.shape_index, pi.instance]
            for g, w in zip(got, want):
                assert np.array_equal(g.cpu().numpy().view(np.uint32), np.asarray(w).view(np.uint32)), (name, naive)
        t = scene.ray_intersect_preliminary(ray, active=active).t.cpu().numpy()
        assert np.isinf(t[~active]).all() and np.isfinite(t[active]).any()
        assert np.array_equal(t[active], scene.ray_intersect_preliminary(ray).t.cpu().numpy()[active])
        maxt2 = np.full(n, 0.7, np.float32)
        want = osc.ray_test_masked(o, dd, maxt2, active)
        for naive in (False, True):
            hit = scene.ray_test(mi.Ray3f(o, dd, maxt2), active=active, naive=naive).cpu().numpy()
            assert np.array_equal(hit, want) and not hit[~active].any() and hit[active].any(), (name, naive)
        # all-false and scalar masks
        assert np.isinf(scene.ray_intersect_preliminary(ray, active=False).t.cpu().numpy()).all()
        assert not scene.ray_test(mi.Ray3f(o, dd, maxt2), active=False).cpu().numpy().any()


def test_ray_intersect_flags_and_mask_vs_oracle(mi, O):
    """Scene::ray_intersect(ray, ray_flags, ..., active): Minimal / Shading / Shading | NormalPartials (+ FollowShape, DetachShape, value-neutral) with a mask"""
    n = 12000
    R = mi.RayFlags
    for name, d in _scenes(mi, O):
        scene = mi.load_dict(d)
        osc, _ = O.scene_from_product(scene)
        o, dd, maxt = _rays(n, 5)
        ray = mi.Ray3f(o, dd, maxt)
        active = np.random.default_rng(2).random(n) < 0.75
        pi = scene.ray_intersect_preliminary(ray)
        t = pi.t.cpu().numpy(); u = pi.prim_uv[0].cpu().numpy(); v = pi.prim_uv[1].cpu().numpy()
        prim = pi.prim_index.cpu().numpy().astype(np.uint32); shape = pi.shape_index.cpu().numpy().astype(np.uint32); inst = pi.instance.cpu().numpy().astype(np.uint32)
        sel = np.concatenate([np.flatnonzero(np.isfinite(t))[:1500], np.flatnonzero(~np.isfinite(t))[:100], np.flatnonzero(~active)[:200]])
        # (NormalPartials without Shading computes nothing, as in mesh.cpp:2334 `if (shading)`)
        for flags in (R.Minimal, R.Shading, R.Shading | R.NormalPartials, R.Shading | R.NormalPartials | R.FollowShape, R.Minimal | R.DetachShape, R.NormalPartials):
            si = scene.ray_intersect(ray, flags, active=active)
            got = np.concatenate([si.p.cpu().numpy(), si.n.cpu().numpy(), si.sh_frame.n.cpu().numpy(), si.sh_frame.s.cpu().numpy(), si.sh_frame.t.cpu().numpy(), si.wi.cpu().numpy(),
                                  si.uv.cpu().numpy(), si.t.cpu().numpy()[None], si.dp_du.cpu().numpy(), si.dp_dv.cpu().numpy(), si.dn_du.cpu().numpy(), si.dn_dv.cpu().numpy()])
            assert got.shape == (33, n)
            want = osc.surface_interaction_flags(o[:, sel], dd[:, sel], t[sel], u[sel], v[sel], prim[sel], shape[sel], inst[sel], flags, active[sel])
            g = got[:, sel]
            assert np.array_equal(np.isinf(g[20]), np.isinf(want[20]))
            fin = np.isfinite(want[20])
            assert np.allclose(g[:, fin], want[:, fin], rtol=3e-6, atol=3e-7), (name, flags, float(np.abs(g[:, fin] - want[:, fin]).max()))
            assert np.array_equal(g[:20, ~fin], want[:20, ~fin]) and np.array_equal(g[21:, ~fin], want[21:, ~fin]), (name, flags)       # misses / masked lanes: zeros, wi = -d
            if not (flags & R.Shading):
                assert (got[6:20] == 0).all() and (got[21:] == 0).all() and np.abs(got[0:6, np.isfinite(got[20])]).max() > 0
            if not (flags & R.NormalPartials) or not (flags & R.Shading):
                assert (got[27:] == 0).all()
            # the PreliminaryIntersection of a masked lane is the miss record
            assert np.isinf(si.t.cpu().numpy()[~active]).all()
        if name != "cornell":       # smooth-shaded spheres: the normal partials are there, and tangent to the shading normal
            si = scene.ray_intersect(ray, R.Shading | R.NormalPartials)
            ok = np.isfinite(si.t.cpu().numpy()); dn = si.dn_du.cpu().numpy()[:, ok]; sn = si.sh_frame.n.cpu().numpy()[:, ok]
            assert np.abs(dn).max() > 1e-2 and np.abs((dn * sn).sum(0)).max() < 1e-3 * np.abs(dn).max()
        # ray_intersect_naive = the same interaction through the brute-force kernel; compute_surface_interaction(ray, flags, active) on its own
        a = scene.ray_intersect(ray); b = scene.ray_intersect_naive(ray)
        assert np.array_equal(a.t.cpu().numpy(), b.t.cpu().numpy()) and np.array_equal(a.p.cpu().numpy(), b.p.cpu().numpy())
        c = pi.compute_surface_interaction(ray, R.Shading | R.NormalPartials, active)
        e = scene.ray_intersect(ray, R.Shading | R.NormalPartials, active=active)
        assert np.array_equal(c.dn_dv.cpu().numpy(), e.dn_dv.cpu().numpy()) and np.array_equal(c.wi.cpu().numpy(), e.wi.cpu().numpy())
    # flags the entry points do not know are refused, never ignored
    for bad in (16, 0x100, R.FollowShape | R.DetachShape):
        with pytest.raises(Exception):
            scene.ray_intersect(ray, bad)


def test_bsdf_context_and_mask_device_vs_oracle(mi, O):
    """BSDF::eval / pdf / eval_pdf / sample(ctx, si, ..., active) of every plugin on the GPU vs the oracle's per-plugin context restatement"""
    from tests.test_bsdfs_cpu import BSDF_DICTS
    from tests.test_bsdf_context_cpu import CONTEXTS
    rng = np.random.default_rng(15); n = 384
    z = rng.uniform(-1, 1, (2, n)); ph = rng.uniform(0, 2 * np.pi, (2, n)); r = np.sqrt(1 - z * z)
    wi = np.stack([r[0] * np.cos(ph[0]), r[0] * np.sin(ph[0]), z[0]]).astype(np.float32)
    wo = np.stack([r[1] * np.cos(ph[1]), r[1] * np.sin(ph[1]), z[1]]).astype(np.float32)
    s1 = rng.random(n).astype(np.float32); s2 = rng.random((2, n)).astype(np.float32)
    active = rng.random(n) < 0.8
    uv = np.zeros((2, n), np.float32)
    types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3, "conductor": 4, "plastic": 5}
    nonzero = 0
    for name, d in BSDF_DICTS.items():
        bsdf = mi.load_dict(d)
        bsdf._bind()                                          # a stand-alone plugin lives in a private scene
        si = type("SI", (), dict(wi=wi, uv=None))()
        sd = O.SceneData()
        sd.bsdfs = [(types[b.kind], -1, b.value, dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta, eta_c=b.eta_c,
                                                     k_c=b.k_c, back=b.back.index if b.back is not None else -1)) for b in bsdf.scene.bsdf_objs]
        osc = O.OracleScene(sd)
        for ctx in CONTEXTS:
            c = mi.BSDFContext(ctx[0], ctx[1], ctx[2])
            val, pdf = bsdf.eval_pdf(c, si, wo, active)
            rv, rp = osc.bsdf_evaluate_ctx(bsdf.index, ctx, 2, wi, uv, wo, active)
            assert np.allclose(val.cpu().numpy(), rv, rtol=2e-6, atol=1e-9) and np.allclose(pdf.cpu().numpy(), rp, rtol=2e-6, atol=1e-9), (name, ctx)
            assert np.array_equal(bsdf.eval(c, si, wo, active).cpu().numpy(), val.cpu().numpy()) and np.array_equal(bsdf.pdf(c, si, wo, active).cpu().numpy(), pdf.cpu().numpy())
            assert np.allclose(osc.bsdf_evaluate_ctx(bsdf.index, ctx, 0, wi, uv, wo, active)[0], rv, rtol=2e-6, atol=1e-9)       # the reference's separate eval() ...
            assert np.allclose(osc.bsdf_evaluate_ctx(bsdf.index, ctx, 1, wi, uv, wo, active)[1], rp, rtol=2e-6, atol=1e-9)       # ... and pdf()
            bs, w = bsdf.sample(c, si, s1, s2, active)
            ref = osc.bsdf_sample_ctx(bsdf.index, ctx, wi, uv, s1, s2, active)
            assert np.array_equal(bs.sampled_type.cpu().numpy().astype(np.uint32), ref["sampled_type"]), (name, ctx)
            assert np.array_equal(bs.sampled_component.cpu().numpy().astype(np.uint32), ref["sampled_component"]), (name, ctx)
            assert np.array_equal(bs.eta.cpu().numpy(), ref["eta"]), (name, ctx)
            assert np.allclose(bs.wo.cpu().numpy(), ref["wo"], rtol=2e-6, atol=1e-7) and np.allclose(bs.pdf.cpu().numpy(), ref["pdf"], rtol=2e-6, atol=1e-9), (name, ctx)
            assert np.allclose(w.cpu().numpy(), ref["weight"], rtol=5e-6, atol=1e-9), (name, ctx)
            m = ~active
            assert (val.cpu().numpy()[:, m] == 0).all() and (pdf.cpu().numpy()[m] == 0).all() and (w.cpu().numpy()[:, m] == 0).all() and (bs.wo.cpu().numpy()[:, m] == 0).all()
            nonzero += int((w.cpu().numpy() > 0).any()) + int((val.cpu().numpy() > 0).any())
        # BSDFContext() == NULL context == the context-free wavefront code (bit for bit)
        v0, p0 = bsdf.eval_pdf(None, si, wo); v1, p1 = bsdf.eval_pdf(mi.BSDFContext(), si, wo)
        assert np.array_equal(v0.cpu().numpy(), v1.cpu().numpy(), equal_nan=True) and np.array_equal(p0.cpu().numpy(), p1.cpu().numpy(), equal_nan=True)
        assert bsdf.component_count() == {"twosided_plastic": 4, "twosided_diffuse": 2, "twosided_pair": 2}.get(name, {"dielectric": 2, "roughplastic": 2, "plastic": 2}.get(bsdf.kind, 1))
    assert nonzero > 150
    with pytest.raises(Exception):
        bsdf.eval(mi.BSDFContext(mode=2), si, wo)


def test_reference_dielectric_context_kats_on_device(mi):
    """src/bsdfs/tests/test_dielectric.py test02-04 (tests/golden/reference_kats.json: dielectric_context) through har_bsdf_sample"""
    k = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))["dielectric_context"]
    bsdf = mi.load_dict({"type": "dielectric", **k["bsdf"]})
    for c in k["cases"]:
        si = type("SI", (), dict(wi=np.float32(c["wi"]).reshape(3, 1), uv=None))()
        bs, w = bsdf.sample(mi.BSDFContext(*c["ctx"]), si, np.float32([c["sample1"]]), np.zeros((2, 1), np.float32))
        assert np.allclose(w.cpu().numpy()[:, 0], c["weight"], rtol=1e-5, atol=1e-7), c
        if c.get("zero_only"):
            continue
        assert np.isclose(bs.pdf.cpu().numpy()[0], c["pdf"], rtol=1e-5) and np.isclose(bs.eta.cpu().numpy()[0], c["eta"], rtol=1e-6), c
        assert np.allclose(bs.wo.cpu().numpy()[:, 0], c["wo"], atol=1e-6) and int(bs.sampled_type[0]) == c["type"] and int(bs.sampled_component[0]) == c["component"], c


@pytest.mark.parametrize("kind", ["path", "prb"])
def test_integrator_sample_mask_vs_oracle(mi, O, kind):
    """SamplingIntegrator::sample(..., active): masked rays return zero / invalid and their sampler streams do not move"""
    n = 40000
    for name, d in _scenes(mi, O):
        d["integrator"] = {"type": kind, "max_depth": 5, "rr_depth": 3}
        scene = mi.load_dict(d)
        osc, _ = O.scene_from_product(scene)
        o, dd, maxt = _rays(n, 13)
        active = np.random.default_rng(4).random(n) < 0.7
        sampler = mi.Sampler({"sample_count": 4, "seed": 2}); sampler.seed(1, n)
        before = sampler.state.cpu().numpy().view(np.uint64).copy()
        spec, valid = scene.integrator().sample(scene, sampler, mi.Ray3f(o, dd, maxt), active=active)
        ref, rvalid, rstate = osc.integrator_sample(o, dd, maxt, seed=2 + 1, max_depth=5, rr_depth=3, prb=(kind == "prb"), active=active)
        spec = spec.cpu().numpy(); valid = valid.cpu().numpy()
        assert np.array_equal(valid.astype(np.uint8), rvalid) and not valid[~active].any(), name
        assert (spec[:, ~active] == 0).all() and rel_l2(spec, ref) < 1e-4, name
        if kind == "path":
            after = sampler.state.cpu().numpy().view(np.uint64)
            assert np.array_equal(after, rstate), name
            assert np.array_equal(after[~active], before[~active]) and (after[active] != before[active]).all(), name
        # the unmasked lanes do not depend on the mask
        s2 = mi.Sampler({"sample_count": 4, "seed": 2}); s2.seed(1, n)
        full, _ = scene.integrator().sample(scene, s2, mi.Ray3f(o, dd, maxt))
        assert np.array_equal(full.cpu().numpy()[:, active], spec[:, active]), name


@pytest.mark.parametrize("mode", [("nearest", "repeat"), ("bilinear", "clamp"), ("bilinear", "mirror"), ("nearest", "mirror")])
def test_bitmap_filter_and_wrap_modes_render_and_gradient(mi, O, mode):
    """BitmapTexture filter_type / wrap_mode (src/textures/bitmap.cpp:182-206) in the kernels: a textured Cornell box whose `white` texture is looked up far outside
    [0, 1] (uv scaled by the mesh's texcoords are in [0, 1]; the texture's own `to_uv` is not part of this path, so the test scales the texcoords of the floor) --
    forward image and PRB texel gradients (direct atomics: the band queues serve bilinear + repeat only) vs the oracle"""
    d = mi.textured_cornell_box(res=40, tex_res=8, spp=16)
    tex = np.random.default_rng(6).uniform(0.2, 0.9, (8, 6, 3)).astype(np.float32)
    d["white"]["reflectance"] = {"type": "bitmap", "data": tex, "raw": True, "filter_type": mode[0], "wrap_mode": mode[1]}
    scene = mi.load_dict(d)
    # stretch the texture coordinates of every mesh that carries `white` to [-1.5, 2.5]^2 so that the wrap mode matters
    for m in scene.meshes:
        if scene.bsdf_objs[m["bsdf"]].id == "white":
            m["V"][:, 6:8] = m["V"][:, 6:8] * 4.0 - 1.5
    scene._h = None                                     # re-lower with the edited records
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=16, seed=3).cpu().numpy()
    ref, _ = osc.render_prb(sensor, seed=3, spp=16, max_depth=6)
    assert rel_l2(img, ref) < 1e-4, mode
    grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (40, 40, 3)).astype(np.float32)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=9, spp=16)
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=9, spp=16, max_depth=6)
    got = grads["white.reflectance.data"].cpu().numpy()
    assert got.shape == tex.shape and np.abs(g_tex[0]).max() > 0
    assert rel_l2(got, g_tex[0]) < 1e-3, mode
