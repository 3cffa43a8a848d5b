"""BSDF models (diffuse, dielectric, roughconductor, roughplastic, twosided) on the CPU: the oracle against the
reference's known answers, internal consistency in the style of the reference's own BSDF tests, the product's
HAR_HD code (host harness) against the oracle, and PRB gradients against finite differences."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

BSDF_DICTS = {
    "diffuse": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.8]}},
    "dielectric": {"type": "dielectric", "specular_reflectance": 0.3, "specular_transmittance": 0.6, "int_ior": 1.5, "ext_ior": 1.0},
    "rc_beckmann": {"type": "roughconductor", "alpha": 0.25, "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
    "rc_ggx_aniso": {"type": "roughconductor", "distribution": "ggx", "alpha_u": 0.1, "alpha_v": 0.3, "specular_reflectance": {"type": "rgb", "value": [0.9, 0.8, 0.7]}},
    "rc_ggx_all": {"type": "roughconductor", "distribution": "ggx", "alpha": 0.3, "sample_visible": False},
    "rc_beckmann_all": {"type": "roughconductor", "alpha_u": 0.2, "alpha_v": 0.35, "sample_visible": False},
    "rp_beckmann": {"type": "roughplastic", "diffuse_reflectance": {"type": "rgb", "value": [0.7, 0.3, 0.1]}, "alpha": 0.15},
    "rp_ggx_nonlinear": {"type": "roughplastic", "distribution": "ggx", "alpha": 0.3, "nonlinear": True, "int_ior": 1.9,
                         "diffuse_reflectance": {"type": "rgb", "value": [0.4, 0.6, 0.2]}, "specular_reflectance": {"type": "rgb", "value": [0.9, 0.9, 0.5]}},
    "twosided_diffuse": {"type": "twosided", "b": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.1, 0.1, 0.1]}}},
    "twosided_pair": {"type": "twosided", "front": {"type": "roughconductor", "alpha": 0.2},
                      "back": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.9, 0.9]}}},
}


@pytest.fixture(scope="module")
def H(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p
    for n in ("hh_scene_destroy", "hh_bsdf_eval_pdf", "hh_bsdf_sample", "hh_roughplastic_tables"):
        getattr(L, n).restype = None
    return L


class Pair:
    """one BSDF plugin instantiated in the oracle and in the product's host build"""

    def __init__(self, mi, O, H, d):
        from tests.test_cpu_host import oracle_scene_from
        self.O, self.H = O, H
        bsdf = mi.load_dict(d)
        self.scene = bsdf._bind and mi.core.Scene({'_bsdf': bsdf}) if bsdf.scene is None else bsdf.scene
        self.index = bsdf.index
        sd = O.SceneData()
        types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3}
        sd.bsdfs = [(types[b.kind], -1, b.value, dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta,
                                                     eta_c=b.eta_c, k_c=b.k_c, back=b.back.index if b.back is not None else -1)) for b in self.scene.bsdf_objs]
        self.osc = O.OracleScene(sd)
        desc = self.scene.desc(); err = C.create_string_buffer(256)
        self.h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert self.h, err.value

    def eval_pdf(self, which, wi, wo):
        O = self.O; wi = O.f32(wi); wo = O.f32(wo); uv = O.f32([0.3, 0.6]); val = np.empty(3, np.float32); pdf = C.c_float()
        if which == "oracle":
            O.lib().orc_bsdf_eval_pdf(self.osc.handle, self.index, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
        else:
            self.H.hh_bsdf_eval_pdf(self.h, self.index, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
        return val.copy(), pdf.value

    def sample(self, which, wi, s1, s2):
        O = self.O; wi = O.f32(wi); s2 = O.f32(s2); uv = O.f32([0.3, 0.6])
        wo = np.empty(3, np.float32); w = np.empty(3, np.float32); pdf = C.c_float(); eta = C.c_float(); delta = C.c_int()
        if which == "oracle":
            O.lib().orc_bsdf_sample(self.osc.handle, self.index, O.fp(wi), O.fp(uv), C.c_float(s1), O.fp(s2), O.fp(wo), C.byref(pdf), O.fp(w), C.byref(eta), C.byref(delta))
        else:
            self.H.hh_bsdf_sample(self.h, self.index, O.fp(wi), O.fp(uv), C.c_float(s1), O.fp(s2), O.fp(wo), C.byref(pdf), O.fp(w), C.byref(eta), C.byref(delta))
        return wo.copy(), pdf.value, w.copy(), eta.value, bool(delta.value)


def _sphere(u, v):
    z = 1 - 2 * u; r = np.sqrt(max(0.0, 1 - z * z)); p = 2 * np.pi * v
    return [r * np.cos(p), r * np.sin(p), z]


def test_reference_dielectric_and_twosided_kats(mi, O, H):
    k = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
    b = k["dielectric_sample"]["bsdf"]
    P = Pair(mi, O, H, {"type": "dielectric", **b})
    for which in ("oracle", "product"):
        for c in k["dielectric_sample"]["cases"]:
            wo, pdf, w, eta, delta = P.sample(which, c["wi"], c["sample1"], [0, 0])
            assert np.allclose(w, c["weight"], rtol=1e-5) and np.isclose(pdf, c["pdf"], rtol=1e-5) and np.isclose(eta, c["eta"], rtol=1e-6), (which, c)
            assert np.allclose(wo, c["wo"], atol=1e-6) and delta == c["delta"]
            assert P.eval_pdf(which, c["wi"], c["wo"])[1] == 0.0                  # delta lobes: eval = pdf = 0
    P = Pair(mi, O, H, BSDF_DICTS["twosided_diffuse"])
    for which in ("oracle", "product"):
        for c in k["twosided_pdf"]["cases"]:
            assert np.isclose(P.eval_pdf(which, k["twosided_pdf"]["wi"], c["wo"])[1], c["pdf"], rtol=1e-6, atol=1e-9)
        v_front, p_front = P.eval_pdf(which, [0.3, 0.1, 0.8], [0.1, -0.2, 0.9])
        v_back, p_back = P.eval_pdf(which, [0.3, 0.1, -0.8], [0.1, -0.2, -0.9])
        assert np.allclose(v_front, v_back) and p_front == p_back and p_front > 0       # same BSDF on both sides (twosided.cpp:124-127)


@pytest.mark.parametrize("name", [n for n in BSDF_DICTS if n != "dielectric"])
def test_sample_eval_pdf_consistency(mi, O, H, name):
    """src/bsdfs/tests/test_twosided.py:66-93 / test_rough_conductor.py:98-117 pattern: weight * pdf == eval, pdf == pdf, no NaNs"""
    P = Pair(mi, O, H, BSDF_DICTS[name])
    n = 5; checked = 0
    for which in ("oracle", "product"):
        for u in range(n):
            for v in range(n):
                wi = _sphere((u + 0.5) / n, v / float(n - 1))
                for x in range(n):
                    for y in range(n):
                        s2 = [(x + 0.37) / n, (y + 0.61) / n]
                        wo, pdf, w, eta, delta = P.sample(which, wi, 0.35, s2)
                        if not (w > 0).any():
                            continue
                        val, p2 = P.eval_pdf(which, wi, wo)
                        assert np.isfinite(val).all() and np.isfinite(w).all() and not delta and eta == 1.0
                        assert np.isclose(pdf, p2, rtol=2e-4), (which, wi, s2, pdf, p2)
                        assert np.allclose(w * pdf, val, rtol=2e-3, atol=1e-6), (which, wi, s2, w * pdf, val)
                        checked += 1
    assert checked > 200


@pytest.mark.parametrize("name", list(BSDF_DICTS))
def test_product_host_code_matches_oracle(mi, O, H, name):
    """the product's HAR_HD BSDF code (compiled for the host) against the independent oracle restatement"""
    P = Pair(mi, O, H, BSDF_DICTS[name])
    rng = np.random.default_rng(11)
    for _ in range(400):
        wi = _sphere(rng.random(), rng.random()); wo = _sphere(rng.random(), rng.random()); s1 = float(rng.random()); s2 = rng.random(2)
        a, b = P.eval_pdf("oracle", wi, wo), P.eval_pdf("product", wi, wo)
        assert np.allclose(a[0], b[0], rtol=2e-5, atol=1e-7) and np.isclose(a[1], b[1], rtol=2e-5, atol=1e-7)
        a, b = P.sample("oracle", wi, s1, s2), P.sample("product", wi, s1, s2)
        assert np.allclose(a[0], b[0], atol=2e-6) and np.isclose(a[1], b[1], rtol=5e-5, atol=1e-7) and np.allclose(a[2], b[2], rtol=5e-5, atol=1e-7)
        assert a[3] == b[3] and a[4] == b[4]


def test_roughplastic_tables(mi, O, H):
    """RoughPlastic::parameters_changed (roughplastic.cpp:204-242): product host lowering == oracle; physical sanity"""
    for name in ("rp_beckmann", "rp_ggx_nonlinear"):
        P = Pair(mi, O, H, BSDF_DICTS[name])
        a = np.empty(66, np.float32); b = np.empty(66, np.float32)
        O.lib().orc_roughplastic_tables(P.osc.handle, P.index, O.fp(a)); H.hh_roughplastic_tables(P.h, P.index, O.fp(b))
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
        t = a[:64]
        assert (t >= 0).all() and (t <= 1).all() and t[-1] > 0.85 and t[1] < t[-1] and 0 < a[64] < 1 and 0 < a[65] < 1


def _material_cbox(mi, res):
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["white"] = dict(BSDF_DICTS["rp_beckmann"])
    d["green"] = {"type": "twosided", "m": dict(BSDF_DICTS["rc_ggx_aniso"])}
    d["red"] = dict(BSDF_DICTS["twosided_pair"])
    d["glass"] = {"type": "dielectric", "int_ior": 1.5}
    d["small-box"]["bsdf"] = {"type": "ref", "id": "glass"}
    return d


@pytest.mark.parametrize("mode,md", [(0, 8), (1, 6)])
def test_material_scene_host_pipeline_matches_oracle(mi, O, mode, md):
    """path and prb primal on a Cornell box with every BSDF type (incl. transmission through the glass box)"""
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    scene = mi.load_dict(_material_cbox(mi, 32))
    osc, sensor = oracle_scene_from(O, scene)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((32, 32, 4), np.float32)
    assert L.hh_render(h, C.byref(sensor), mode, 3, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, st = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=8, max_depth=md, raw=True, threads=2)
    assert np.isfinite(film).all() and rel_l2(O.develop(film), O.develop(ref)) < 1e-4


def test_oracle_prb_gradients_vs_finite_differences_materials(mi, O):
    """slot-0 colour parameters of roughplastic (diffuse_reflectance) and roughconductor (specular_reflectance)"""
    from tests.test_cpu_host import oracle_scene_from
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 12; d["sensor"]["film"]["height"] = 12
    d["white"] = dict(BSDF_DICTS["rp_beckmann"]); d["green"] = dict(BSDF_DICTS["rc_ggx_aniso"])
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    # 2048 spp: perturbing roughplastic's diffuse_reflectance also moves its lobe-selection probability
    # (specular_sampling_weight), so the finite difference of the SAME-seed estimator is noisier than for the other plugins
    seed, spp, md = 5, 2048, 4
    grad_in = np.ones((12, 12, 3), np.float32)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=seed, spp=spp, max_depth=md)
    for bsdf, chan, base in ((0, 0, scene.bsdf_objs[0].value), (1, 1, scene.bsdf_objs[1].value)):
        eps = 2e-2; sums = []
        for sgn in (+1, -1):
            v = np.array(base, np.float32); v[chan] += sgn * eps; osc.set_reflectance(bsdf, v)
            img, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=md); sums.append(img.astype(np.float64).sum())
        osc.set_reflectance(bsdf, np.array(base, np.float32))
        fd = (sums[0] - sums[1]) / (2 * eps)
        assert abs(fd - g_refl[bsdf, chan]) / abs(fd) < 1.5e-2, (bsdf, fd, g_refl[bsdf, chan])
