"""BSDFContext (mode / type_mask / component) and the Mask argument of BSDF::eval / pdf / eval_pdf / sample on the CPU:
the reference's own dielectric known answers (src/bsdfs/tests/test_dielectric.py:30-136) against the oracle AND the product's
host build, the oracle's per-plugin context restatement (oracle/orc_bsdf_ctx.h) pinned to the golden-vector-pinned default-context
oracle, and the product's HAR_HD context code against that restatement for every model, twosided wrapper, context and mask."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests.test_bsdfs_cpu import BSDF_DICTS, Pair, _sphere

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ALL, NONE = 0x1ff, 0xffffffff
DIFFUSE_R, GLOSSY_R, DELTA_R, DELTA_T = 0x2, 0x8, 0x20, 0x40

CONTEXTS = [(0, ALL, NONE), (1, ALL, NONE), (0, NONE, NONE), (0, ALL, 0), (0, ALL, 1), (0, ALL, 2), (0, ALL, 3), (1, ALL, 1),
            (0, DIFFUSE_R, NONE), (0, GLOSSY_R, NONE), (0, DELTA_R, NONE), (0, DELTA_T, NONE), (0, DIFFUSE_R | GLOSSY_R, NONE), (0, DELTA_R | DELTA_T, NONE),
            (0, DELTA_R | DIFFUSE_R, 1), (0, 0, NONE), (1, GLOSSY_R, 0)]


@pytest.fixture(scope="module")
def H(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p
    for n in ("hh_scene_destroy", "hh_bsdf_eval_pdf", "hh_bsdf_sample", "hh_bsdf_eval_pdf_ctx", "hh_bsdf_sample_ctx", "hh_roughplastic_tables"):
        getattr(L, n).restype = None
    return L


def product_eval(P, ctx, wi, wo, active=True):
    O = P.O; val = np.empty(3, np.float32); pdf = C.c_float()
    P.H.hh_bsdf_eval_pdf_ctx(P.h, P.index, C.c_uint32(ctx[0]), C.c_uint32(ctx[1]), C.c_uint32(ctx[2]), 1 if active else 0, O.fp(O.f32(wi)), O.fp(O.f32([0.3, 0.6])), O.fp(O.f32(wo)),
                             O.fp(val), C.byref(pdf))
    return val.copy(), pdf.value


def product_sample(P, ctx, wi, s1, s2, active=True):
    O = P.O; wo = np.empty(3, np.float32); w = np.empty(3, np.float32); pdf = C.c_float(); eta = C.c_float(); st = C.c_uint32(); sc = C.c_uint32()
    P.H.hh_bsdf_sample_ctx(P.h, P.index, C.c_uint32(ctx[0]), C.c_uint32(ctx[1]), C.c_uint32(ctx[2]), 1 if active else 0, O.fp(O.f32(wi)), O.fp(O.f32([0.3, 0.6])), C.c_float(s1),
                           O.fp(O.f32(s2)), O.fp(wo), C.byref(pdf), O.fp(w), C.byref(eta), C.byref(st), C.byref(sc))
    return dict(wo=wo.copy(), pdf=pdf.value, weight=w.copy(), eta=eta.value, sampled_type=st.value, sampled_component=sc.value)


def oracle_eval(P, ctx, which, wi, wo, active=True):
    v, p = P.osc.bsdf_evaluate_ctx(P.index, ctx, which, np.reshape(wi, (3, 1)), np.reshape([0.3, 0.6], (2, 1)), np.reshape(wo, (3, 1)), None if active else [0])
    return v[:, 0], float(p[0])


def oracle_sample(P, ctx, wi, s1, s2, active=True):
    r = P.osc.bsdf_sample_ctx(P.index, ctx, np.reshape(wi, (3, 1)), np.reshape([0.3, 0.6], (2, 1)), [s1], np.reshape(s2, (2, 1)), None if active else [0])
    return dict(wo=r["wo"][:, 0], pdf=float(r["pdf"][0]), weight=r["weight"][:, 0], eta=float(r["eta"][0]), sampled_type=int(r["sampled_type"][0]),
                sampled_component=int(r["sampled_component"][0]))


def test_reference_dielectric_context_kats(mi, O, H):
    """src/bsdfs/tests/test_dielectric.py test02 / test03 (both transport modes) and test04_sample_specific_component, transcribed by tests/golden/make_golden.py"""
    k = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))["dielectric_context"]
    P = Pair(mi, O, H, {"type": "dielectric", **k["bsdf"]})
    assert len(k["cases"]) == 33
    for which, fn in (("oracle", oracle_sample), ("product", product_sample)):
        for c in k["cases"]:
            r = fn(P, c["ctx"], c["wi"], c["sample1"], [0, 0])
            assert np.allclose(r["weight"], c["weight"], rtol=1e-5, atol=1e-7), (which, c, r)
            if c.get("zero_only"):
                assert (r["weight"] == 0).all()
                continue
            assert np.isclose(r["pdf"], c["pdf"], rtol=1e-5) and np.isclose(r["eta"], c["eta"], rtol=1e-6), (which, c, r)
            assert np.allclose(r["wo"], c["wo"], atol=1e-6) and r["sampled_type"] == c["type"] and r["sampled_component"] == c["component"], (which, c, r)


@pytest.mark.parametrize("name", list(BSDF_DICTS))
def test_context_oracle_is_pinned_to_the_default_context_oracle(mi, O, H, name):
    """orc_bsdf_ctx.h under BSDFContext() must reproduce orc_bsdf.h (which the reference's golden vectors pin): eval / pdf / eval_pdf / sample, bit for bit"""
    P = Pair(mi, O, H, BSDF_DICTS[name])
    rng = np.random.default_rng(5)
    for _ in range(300):
        wi = _sphere(rng.random(), rng.random()); wo = _sphere(rng.random(), rng.random()); s1 = float(rng.random()); s2 = rng.random(2)
        v0, p0 = P.eval_pdf("oracle", wi, wo)
        for which in (0, 1, 2):
            v, p = oracle_eval(P, (0, ALL, NONE), which, wi, wo)
            if which != 1:
                assert np.array_equal(v, v0), (name, which, wi, wo, v, v0)
            if which != 0:
                assert p == p0, (name, which, wi, wo, p, p0)
        wo0, pdf0, w0, eta0, delta0 = P.sample("oracle", wi, s1, s2)
        r = oracle_sample(P, (0, ALL, NONE), wi, s1, s2)
        assert np.array_equal(r["wo"], wo0) and r["pdf"] == pdf0 and np.array_equal(r["weight"], w0) and r["eta"] == eta0, (name, wi, s1, s2, r, wo0, pdf0, w0)
        assert ((r["sampled_type"] & (DELTA_R | DELTA_T)) != 0) == delta0


@pytest.mark.parametrize("name", list(BSDF_DICTS))
def test_product_context_code_matches_oracle(mi, O, H, name):
    """the product's CTX = true instantiations (har_bsdf.h, bsdf_side_ctx) against the per-plugin restatement, for every context and for masked lanes"""
    P = Pair(mi, O, H, BSDF_DICTS[name])
    rng = np.random.default_rng(23)
    nonzero = 0
    for ctx in CONTEXTS:
        for k in range(60):
            wi = _sphere(rng.random(), rng.random()); wo = _sphere(rng.random(), rng.random()); s1 = float(rng.random()); s2 = rng.random(2)
            active = k % 7 != 3
            pv, pp = product_eval(P, ctx, wi, wo, active)
            ov, op = oracle_eval(P, ctx, 2, wi, wo, active)
            assert np.allclose(pv, ov, rtol=2e-6, atol=1e-9) and np.isclose(pp, op, rtol=2e-6, atol=1e-9), (name, ctx, wi, wo, pv, ov, pp, op)
            # BSDF::eval and ::pdf alone are the halves of the pair (three separate restatements on the oracle side)
            assert np.allclose(oracle_eval(P, ctx, 0, wi, wo, active)[0], pv, rtol=2e-6, atol=1e-9) and np.isclose(oracle_eval(P, ctx, 1, wi, wo, active)[1], pp, rtol=2e-6, atol=1e-9)
            ps = product_sample(P, ctx, wi, s1, s2, active); osm = oracle_sample(P, ctx, wi, s1, s2, active)
            assert ps["sampled_type"] == osm["sampled_type"] and ps["sampled_component"] == osm["sampled_component"], (name, ctx, wi, s1, s2, ps, osm)
            assert np.allclose(ps["wo"], osm["wo"], rtol=2e-6, atol=1e-7) and np.isclose(ps["pdf"], osm["pdf"], rtol=2e-6, atol=1e-9) and ps["eta"] == osm["eta"], (name, ctx, ps, osm)
            assert np.allclose(ps["weight"], osm["weight"], rtol=5e-6, atol=1e-9), (name, ctx, wi, s1, s2, ps, osm)
            if not active:
                assert (pv == 0).all() and pp == 0 and (ps["weight"] == 0).all() and ps["pdf"] == 0 and (ps["wo"] == 0).all() and ps["eta"] == 0 and ps["sampled_type"] == 0
            nonzero += int((ps["weight"] > 0).any()) + int((pv > 0).any())
    assert nonzero > 50


def test_context_semantics_by_construction(mi, O, H):
    """properties the reference's sources imply, checked on the product's host code"""
    # roughplastic: the two single-component evaluations add up to the full one; a single-lobe context samples that lobe with certainty
    P = Pair(mi, O, H, BSDF_DICTS["rp_ggx_nonlinear"])
    rng = np.random.default_rng(3)
    for _ in range(100):
        wi = _sphere(rng.random() * 0.45, rng.random()); wo = _sphere(rng.random() * 0.45, rng.random())
        full, _ = product_eval(P, (0, ALL, NONE), wi, wo); spec, ps = product_eval(P, (0, ALL, 0), wi, wo); diff, pd = product_eval(P, (0, ALL, 1), wi, wo)
        assert np.allclose(spec + diff, full, rtol=1e-5, atol=1e-8)
        assert np.array_equal(product_eval(P, (0, GLOSSY_R, NONE), wi, wo)[0], spec) and np.array_equal(product_eval(P, (0, DIFFUSE_R, NONE), wi, wo)[0], diff)
        assert np.isclose(pd, wo[2] / np.pi, rtol=1e-5)                                    # prob_diffuse = 1 (roughplastic.cpp:403-407)
        for s1 in (0.0, 0.999):
            assert product_sample(P, (0, ALL, 0), wi, s1, rng.random(2))["sampled_component"] == 0
            assert product_sample(P, (0, ALL, 1), wi, s1, rng.random(2))["sampled_component"] == 1
    # plastic: only the delta coating -> weight = f_i * specular_reflectance, pdf 1 (plastic.cpp:231-247); only the base -> pdf = cosine hemisphere (:339-348)
    P = Pair(mi, O, H, BSDF_DICTS["plastic_nonlinear"])
    r = product_sample(P, (0, DELTA_R, NONE), [0.3, 0.2, 0.8], 0.9, [0.4, 0.6])
    fr = np.empty(4, np.float32); O.lib().orc_fresnel.argtypes = [C.c_float, C.c_float, O.c_f32p]; O.lib().orc_fresnel(C.c_float(0.8), C.c_float(np.float32(1.49) / np.float32(1.000277)), O.fp(fr))
    assert r["pdf"] == 1.0 and r["sampled_type"] == DELTA_R and np.allclose(r["weight"], fr[0] * np.array([0.7, 0.9, 0.5]), rtol=1e-5)
    v, p = product_eval(P, (0, DIFFUSE_R, NONE), [0.3, 0.2, 0.8], [0.1, 0.5, 0.7])
    assert np.isclose(p, 0.7 / np.pi, rtol=1e-6) and (v > 0).all()
    assert product_eval(P, (0, DELTA_R, NONE), [0.3, 0.2, 0.8], [0.1, 0.5, 0.7]) == (pytest.approx([0, 0, 0]), 0.0)
    # twosided(front, back): component indices run over the front's lobes, then the back's (twosided.cpp:86-99,129-146)
    P = Pair(mi, O, H, BSDF_DICTS["twosided_pair"])            # front roughconductor (1 lobe), back diffuse (1 lobe)
    wi_f, wi_b = [0.2, 0.1, 0.9], [0.2, 0.1, -0.9]
    assert (product_eval(P, (0, ALL, 0), wi_f, [0.1, 0.1, 0.95])[0] > 0).all() and (product_eval(P, (0, ALL, 1), wi_f, [0.1, 0.1, 0.95])[0] == 0).all()
    assert (product_eval(P, (0, ALL, 1), wi_b, [0.1, 0.1, -0.95])[0] > 0).all() and (product_eval(P, (0, ALL, 2), wi_b, [0.1, 0.1, -0.95])[0] == 0).all()
    # the reference's quirk: on the back side `component - count(front)` wraps in uint32, so component = count(front) - 1 becomes "all" there
    assert (product_eval(P, (0, ALL, 0), wi_b, [0.1, 0.1, -0.95])[0] > 0).all()
    assert (product_eval(P, (0, GLOSSY_R, NONE), wi_b, [0.1, 0.1, -0.95])[0] == 0).all() and (product_eval(P, (0, DIFFUSE_R, NONE), wi_f, [0.1, 0.1, 0.95])[0] == 0).all()
    # one nested BSDF: both sides see the caller's component unchanged, so the "back" indices never match (twosided.cpp:122-127)
    P = Pair(mi, O, H, BSDF_DICTS["twosided_diffuse"])
    assert (product_eval(P, (0, ALL, 0), wi_b, [0.1, 0.1, -0.95])[0] > 0).all() and (product_eval(P, (0, ALL, 1), wi_b, [0.1, 0.1, -0.95])[0] == 0).all()
