"""BSDF models (diffuse, dielectric, conductor, plastic, roughconductor, roughplastic, twosided) on the CPU: the oracle against the
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
    "conductor_mirror": {"type": "conductor"},
    "conductor_gold": {"type": "conductor", "eta": [0.143, 0.375, 1.442], "k": [3.983, 2.386, 1.603], "specular_reflectance": {"type": "rgb", "value": [0.9, 0.8, 0.95]}},
    "plastic": {"type": "plastic", "diffuse_reflectance": {"type": "rgb", "value": [0.1, 0.27, 0.36]}, "int_ior": 1.9},
    "plastic_nonlinear": {"type": "plastic", "nonlinear": True, "diffuse_reflectance": {"type": "rgb", "value": [0.8, 0.6, 0.3]},
                          "specular_reflectance": {"type": "rgb", "value": [0.7, 0.9, 0.5]}},
    "twosided_plastic": {"type": "twosided", "m": {"type": "plastic"}},
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
        types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3, "conductor": 4, "plastic": 5}
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


DELTA_ONLY = ("dielectric", "conductor_mirror", "conductor_gold")


@pytest.mark.parametrize("name", [n for n in BSDF_DICTS if n not in DELTA_ONLY])
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
                        if name in ("plastic", "plastic_nonlinear", "twosided_plastic") and delta:       # plastic's delta lobe: the mirror direction
                            assert np.allclose(wo, [-wi[0], -wi[1], wi[2]], atol=1e-6)
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


def _fresnel_conductor_f64(c, eta, k):
    """unpolarised reflectance of a conductor from the complex Fresnel equations, evaluated independently in complex128"""
    n = eta + 1j * k
    ct = np.sqrt(1 - (1 - c * c) / (n * n))
    rs = (c - n * ct) / (c + n * ct); rp = (n * c - ct) / (n * c + ct)
    return 0.5 * (abs(rs) ** 2 + abs(rp) ** 2)


def _fresnel_dielectric_f64(c, eta):
    s2 = (1 - c * c) / (eta * eta)
    if s2 >= 1:
        return 1.0
    ct = np.sqrt(1 - s2)
    rs = (c - eta * ct) / (c + eta * ct); rp = (eta * c - ct) / (eta * c + ct)
    return 0.5 * (rs * rs + rp * rp)


@pytest.mark.parametrize("name", ["conductor_mirror", "conductor_gold"])
def test_conductor_is_a_fresnel_weighted_mirror(mi, O, H, name):
    """SmoothConductor::sample (src/bsdfs/conductor.cpp:246-304): wo = reflect(wi), pdf = 1, weight = specular_reflectance * F(cos_i), eval = pdf = 0;
    src/bsdfs/tests/test_conductor.py:57-60 checks the same Fresnel identity (through the Mueller matrix) at 45 degrees"""
    d = BSDF_DICTS[name]; P = Pair(mi, O, H, d)
    eta = np.array(d.get("eta", [0, 0, 0]), np.float64); k = np.array(d.get("k", [1, 1, 1]), np.float64)
    refl = np.array(d.get("specular_reflectance", {"value": [1, 1, 1]})["value"], np.float64)
    for which in ("oracle", "product"):
        for theta in (0.0, 20.0, 45.0, 70.0, 89.0):
            t = np.radians(theta); wi = [-np.sin(t), 0.0, np.cos(t)]
            wo, pdf, w, e, delta = P.sample(which, wi, 0.3, [0.2, 0.9])
            assert np.allclose(wo, [np.sin(t), 0, np.cos(t)], atol=1e-6) and pdf == 1.0 and e == 1.0 and delta
            expect = refl * np.array([_fresnel_conductor_f64(np.cos(t), eta[c], k[c]) for c in range(3)])
            assert np.allclose(w, expect, rtol=2e-5), (which, theta, w, expect)
            val, p = P.eval_pdf(which, wi, wo); assert (val == 0).all() and p == 0.0
        wo, pdf, w, e, delta = P.sample(which, [0.3, 0.2, -0.9], 0.3, [0.2, 0.9])          # back side: FrontSide only
        assert pdf == 0.0 and (w == 0).all()
    if name == "conductor_mirror":                  # (eta, k) = (0, 1): the "100% reflecting mirror" of the plugin documentation
        assert np.allclose(P.sample("product", [0.6, 0, 0.8], 0.5, [0.5, 0.5])[2], 1.0, atol=1e-6)


def test_fresnel_diffuse_reflectance_fit_vs_quadrature(mi, O, H):
    """fresnel_diffuse_reflectance (include/mitsuba/render/fresnel.h:327-355) is a polynomial fit of 2 * int_0^1 F(mu, eta) mu dmu;
    the plastic model stores it for 1 / eta (m_fdr_int).  The oracle's value must sit within the fit's accuracy of the quadrature."""
    mu, wq = np.polynomial.legendre.leggauss(400); mu = 0.5 * (mu + 1); wq = 0.5 * wq
    for name, eta in (("plastic", 1.9 / 1.000277), ("plastic_nonlinear", 1.49 / 1.000277)):
        P = Pair(mi, O, H, BSDF_DICTS[name])
        quad = 2.0 * sum(w * _fresnel_dielectric_f64(m, 1.0 / eta) * m for m, w in zip(mu, wq))
        a = np.empty(66, np.float32); b = np.empty(66, np.float32)
        O.lib().orc_roughplastic_tables(P.osc.handle, P.index, O.fp(a)); H.hh_roughplastic_tables(P.h, P.index, O.fp(b))
        assert a[64] == b[64] and a[65] == b[65]                                 # internal_reflectance, specular_sampling_weight: bit-equal
        assert abs(a[64] - quad) < 8e-3, (name, a[64], quad)


@pytest.mark.parametrize("name", ["plastic", "plastic_nonlinear"])
def test_plastic_lobes_and_energy(mi, O, H, name):
    """SmoothPlastic (src/bsdfs/plastic.cpp:208-352): lobe selection probabilities, the delta lobe's weight, the diffuse lobe against
    eval / pdf, pdf normalisation (integrates to the diffuse selection probability) and energy conservation"""
    d = BSDF_DICTS[name]; P = Pair(mi, O, H, d)
    eta = float(np.float32(d.get("int_ior", 1.49)) / np.float32(1.000277))
    spec = np.array(d.get("specular_reflectance", {"value": [1, 1, 1]})["value"], np.float64)
    rho = np.array(d["diffuse_reflectance"]["value"], np.float64)
    ssw = spec.mean() / (rho.mean() + spec.mean())
    rng = np.random.default_rng(5)
    for which in ("oracle", "product"):
        for theta in (0.0, 40.0, 75.0, 88.0):
            t = np.radians(theta); wi = [np.sin(t), 0.0, np.cos(t)]
            f_i = _fresnel_dielectric_f64(np.cos(t), eta)
            ps = f_i * ssw / (f_i * ssw + (1 - f_i) * (1 - ssw))
            wo, pdf, w, e, delta = P.sample(which, wi, ps * 0.999, [0.3, 0.3])              # just below the threshold: specular
            assert delta and np.isclose(pdf, ps, rtol=1e-4) and np.allclose(wo, [-wi[0], 0, wi[2]], atol=1e-6) and e == 1.0
            assert np.allclose(w, spec * f_i / ps, rtol=1e-4)
            wo, pdf, w, e, delta = P.sample(which, wi, min(ps * 1.001 + 1e-6, 0.99999), [0.3, 0.3])   # just above: diffuse
            assert not delta and wo[2] > 0 and e == 1.0
            val, p2 = P.eval_pdf(which, wi, wo)
            assert np.isclose(pdf, p2, rtol=1e-4) and np.allclose(w * pdf, val, rtol=1e-4, atol=1e-7)
            # pdf integrates to the probability of the diffuse lobe; albedo = E[weight] <= 1
            n = 4000; acc_pdf = 0.0; acc_w = np.zeros(3)
            for _ in range(n):
                u = rng.random(3)
                wo_u = _sphere(u[0] * 0.5, u[1])                                           # uniform on the upper hemisphere, pdf 1 / (2 pi)
                acc_pdf += P.eval_pdf(which, wi, wo_u)[1] * 2 * np.pi
                acc_w += P.sample(which, wi, float(u[2]), rng.random(2))[2]
            assert abs(acc_pdf / n - (1 - ps)) < 0.03, (which, theta, acc_pdf / n, 1 - ps)
            assert (acc_w / n <= 1.0 + 1e-3).all(), (which, theta, acc_w / n)
        assert P.sample(which, [0.1, 0.2, -0.9], 0.5, [0.5, 0.5])[1] == 0.0                  # FrontSide only
        assert P.eval_pdf(which, [0.1, 0.2, 0.9], [0.1, 0.2, -0.9])[1] == 0.0


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


def _smooth_material_cbox(mi, res):
    """the smooth counterparts: plastic walls, a gold `conductor` wall, a mirror box"""
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["white"] = dict(BSDF_DICTS["plastic"], diffuse_reflectance={"type": "rgb", "value": [0.7, 0.7, 0.65]})
    d["green"] = {"type": "twosided", "m": dict(BSDF_DICTS["conductor_gold"])}
    d["red"] = dict(BSDF_DICTS["plastic_nonlinear"])
    d["mirror"] = dict(BSDF_DICTS["conductor_mirror"])
    d["large-box"]["bsdf"] = {"type": "ref", "id": "mirror"}
    return d


@pytest.mark.parametrize("mode,md,which", [(0, 8, "rough"), (1, 6, "rough"), (0, 8, "smooth"), (1, 6, "smooth")])
def test_material_scene_host_pipeline_matches_oracle(mi, O, mode, md, which):
    """path and prb primal on a Cornell box with every BSDF type (incl. transmission through the glass box)"""
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    scene = mi.load_dict((_material_cbox if which == "rough" else _smooth_material_cbox)(mi, 32))
    osc, sensor = oracle_scene_from(O, scene)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((32, 32, 4), np.float32)
    assert L.hh_render(h, C.byref(sensor), mode, 3, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, st = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=8, max_depth=md, raw=True, threads=2)
    assert np.isfinite(film).all() and rel_l2(O.develop(film), O.develop(ref)) < 1e-4


def test_oracle_prb_gradients_vs_finite_differences_plastic(mi, O):
    """slot-0 (diffuse_reflectance) of `plastic`, linear and nonlinear.  With a specular lobe PRB is NOT the derivative of the estimator:
    prb.py:288-297 takes relative_grad(bsdf.eval(si, wo)) at the sampled wo whatever lobe produced it, and SmoothPlastic::eval returns the
    diffuse term at the mirror direction too (measured here: +14% on the white walls).  specular_reflectance = 0 never selects that lobe
    (specular_sampling_weight = 0), which isolates the diffuse derivative, the part that has an exact answer."""
    from tests.test_cpu_host import oracle_scene_from
    d = _smooth_material_cbox(mi, 12)
    for k in ("white", "red"):
        d[k]["specular_reflectance"] = {"type": "rgb", "value": [0.0, 0.0, 0.0]}
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    seed, spp, md = 5, 2048, 4
    grad_in = np.ones((12, 12, 3), np.float32)
    g_refl, _, _ = osc.render_prb_backward(sensor, grad_in, seed=seed, spp=spp, max_depth=md)
    names = [b.id for b in scene.bsdf_objs]
    for key, chan in (("white", 0), ("red", 1)):
        bsdf = names.index(key); base = scene.bsdf_objs[bsdf].value
        eps = 2e-2; sums = []
        for sgn in (+1, -1):
            v = np.array(base, np.float32); v[chan] += sgn * eps; osc.set_reflectance(bsdf, v)
            img, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=md); sums.append(img.astype(np.float64).sum())
        osc.set_reflectance(bsdf, np.array(base, np.float32))
        fd = (sums[0] - sums[1]) / (2 * eps)
        assert abs(fd - g_refl[bsdf, chan]) / abs(fd) < 1.5e-2, (key, fd, g_refl[bsdf, chan])


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


def test_bitmap_bilinear_repeat_vs_numpy(mi, O, H):
    """BitmapTexture::eval for a raw RGB bitmap (bitmap.cpp:834-850 -> dr::Texture<Float, 2>::eval, bilinear + repeat; texel centres at (i + 0.5) / res):
    the diffuse BSDF's value * pi / cos(theta_o) at uv is the interpolated texel.  NON-square 7 x 3 bitmap, uv far outside [0, 1] (negative, > 2) and
    exactly on texel centres / seams -- oracle and host-compiled product against an independent NumPy lookup"""
    rng = np.random.default_rng(21)
    Hh, Ww = 3, 7
    tex = rng.uniform(0.1, 0.9, (Hh, Ww, 3)).astype(np.float32)
    P = Pair(mi, O, H, {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True}})
    # Pair builds the oracle scene without textures: give it the bitmap
    sd = O.SceneData(); b = P.scene.bsdf_objs[P.index]
    sd.bsdfs = [(0, 0, b.value, dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta, eta_c=b.eta_c, k_c=b.k_c, back=-1))]
    sd.textures = [tex]
    osc = O.OracleScene(sd)

    def numpy_lookup(u, v):
        px, py = np.float32(u) * Ww - np.float32(0.5), np.float32(v) * Hh - np.float32(0.5)
        x0, y0 = int(np.floor(px)), int(np.floor(py)); fx, fy = float(px - x0), float(py - y0)
        t = lambda x, y: tex[y % Hh, x % Ww].astype(np.float64)
        return (1 - fy) * ((1 - fx) * t(x0, y0) + fx * t(x0 + 1, y0)) + fy * ((1 - fx) * t(x0, y0 + 1) + fx * t(x0 + 1, y0 + 1))

    wi = O.f32([0.2, -0.1, 0.9]); wo = O.f32([-0.3, 0.2, 0.8]); wo /= np.linalg.norm(wo)
    uvs = [(0.5 / Ww, 0.5 / Hh), (0.0, 0.0), (1.0, 1.0), (-0.25, 2.75), (3.9, -1.4), (1.0 - 1e-7, 0.5), ((Ww - 0.5) / Ww, (Hh - 0.5) / Hh)]
    uvs += [tuple(rng.uniform(-3, 4, 2)) for _ in range(200)]
    for u, v in uvs:
        uv = O.f32([u, v]); want = numpy_lookup(u, v)
        for which in ("oracle", "product"):
            val = np.empty(3, np.float32); pdf = C.c_float()
            if which == "oracle":
                O.lib().orc_bsdf_eval_pdf(osc.handle, 0, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
            else:
                H.hh_bsdf_eval_pdf(P.h, P.index, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
            got = val.astype(np.float64) * np.pi / float(wo[2])
            assert np.allclose(got, want, rtol=2e-5, atol=2e-6), (which, u, v, got, want)


@pytest.mark.parametrize("filter_type", ["bilinear", "nearest"])
@pytest.mark.parametrize("wrap_mode", ["repeat", "mirror", "clamp"])
def test_bitmap_filter_and_wrap_modes_vs_numpy(mi, O, H, filter_type, wrap_mode):
    """BitmapTexture `filter_type` / `wrap_mode` (src/textures/bitmap.cpp:182-206 -> dr::FilterMode / dr::WrapMode): oracle and host-compiled product against an
    independent NumPy lookup built on np.pad ('wrap' = repeat, 'symmetric' = mirror: the edge texel repeats, 'edge' = clamp), uv far outside [0, 1]"""
    rng = np.random.default_rng(31)
    Hh, Ww, R = 4, 5, 6                         # R repetitions of padding on every side
    tex = rng.uniform(0.1, 0.9, (Hh, Ww, 3)).astype(np.float32)
    P = Pair(mi, O, H, {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True, "filter_type": filter_type, "wrap_mode": wrap_mode}})
    assert P.scene.texture_modes == [(1 if filter_type == "nearest" else 0) | {"repeat": 0, "mirror": 2, "clamp": 4}[wrap_mode]]
    sd = O.SceneData(); b = P.scene.bsdf_objs[P.index]
    sd.bsdfs = [(0, 0, b.value, dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta, eta_c=b.eta_c, k_c=b.k_c, back=-1))]
    sd.textures = [tex]; sd.texture_modes = list(P.scene.texture_modes)
    osc = O.OracleScene(sd)
    big = np.pad(tex, ((R * Hh, R * Hh), (R * Ww, R * Ww), (0, 0)), mode={"repeat": "wrap", "mirror": "symmetric", "clamp": "edge"}[wrap_mode]).astype(np.float64)

    def numpy_lookup(u, v):
        t = lambda x, y: big[y + R * Hh, x + R * Ww]
        if filter_type == "nearest":
            return t(int(np.floor(np.float32(u) * np.float32(Ww))), int(np.floor(np.float32(v) * np.float32(Hh))))
        px, py = np.float32(u) * Ww - np.float32(0.5), np.float32(v) * Hh - np.float32(0.5)
        x0, y0 = int(np.floor(px)), int(np.floor(py)); fx, fy = float(px - x0), float(py - y0)
        return (1 - fy) * ((1 - fx) * t(x0, y0) + fx * t(x0 + 1, y0)) + fy * ((1 - fx) * t(x0, y0 + 1) + fx * t(x0 + 1, y0 + 1))

    wi = O.f32([0.2, -0.1, 0.9]); wo = O.f32([-0.3, 0.2, 0.8]); wo /= np.linalg.norm(wo)
    uvs = [(0.5 / Ww, 0.5 / Hh), (0.0, 0.0), (1.0, 1.0), (-0.001, 1.001), (-1.0, 2.0), (-0.25, 2.75), (3.9, -1.4), (-4.99, 4.99), ((Ww - 0.5) / Ww, (Hh - 0.5) / Hh)]
    uvs += [tuple(rng.uniform(-5, 6, 2)) for _ in range(300)]
    for u, v in uvs:
        uv = O.f32([u, v]); want = numpy_lookup(u, v)
        for which in ("oracle", "product"):
            val = np.empty(3, np.float32); pdf = C.c_float()
            if which == "oracle":
                O.lib().orc_bsdf_eval_pdf(osc.handle, 0, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
            else:
                H.hh_bsdf_eval_pdf(P.h, P.index, O.fp(wi), O.fp(uv), O.fp(wo), O.fp(val), C.byref(pdf))
            got = val.astype(np.float64) * np.pi / float(wo[2])
            assert np.allclose(got, want, rtol=2e-5, atol=2e-6), (which, filter_type, wrap_mode, u, v, got, want)
    for bad in ({"filter_type": "trilinear"}, {"wrap_mode": "border"}):
        with pytest.raises(RuntimeError, match="Invalid"):
            mi.load_dict({"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, **bad}})


def test_reference_gauss_legendre_known_answers(O, H):
    """src/core/tests/test_quad.py:16-22 (test02_gauss_legendre) for the rule behind rough plastic's transmittance tables (quad.h:27-90), oracle and
    product host code; plus NumPy's leggauss at the table resolution"""
    s = np.sqrt
    kats = {1: ([0.0], [2.0]), 2: ([-s(1 / 3), s(1 / 3)], [1.0, 1.0]), 3: ([-s(3 / 5), 0.0, s(3 / 5)], [5 / 9, 8 / 9, 5 / 9]),
            4: ([-0.861136, -0.339981, 0.339981, 0.861136], [0.347855, 0.652145, 0.652145, 0.347855])}
    H.hh_gauss_legendre.restype = None
    for fn in (O.lib().orc_gauss_legendre, H.hh_gauss_legendre):
        for n, (nodes, weights) in kats.items():
            a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
            fn(n, O.fp(a), O.fp(b))
            assert np.allclose(a, nodes, atol=2e-6) and np.allclose(b, weights, atol=2e-6), (n, a, b)
        for n in (32, 100):
            a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
            fn(n, O.fp(a), O.fp(b))
            x, w = np.polynomial.legendre.leggauss(n)
            assert np.allclose(a, x, atol=3e-6) and np.allclose(b, w, atol=3e-6)


@pytest.mark.parametrize("name", ["diffuse", "rc_beckmann", "rc_ggx_aniso", "rc_ggx_all", "rc_beckmann_all", "rp_beckmann", "rp_ggx_nonlinear", "twosided_pair"])
@pytest.mark.parametrize("which", ["oracle", "product"])
def test_sampling_density_chi2(mi, O, H, name, which):
    """The reference's chi^2 test of BSDF sampling (mitsuba.chi2.ChiSquareTest with BSDFAdapter: src/bsdfs/tests/test_rough_conductor.py:8-95,
    test_rough_plastic.py:6-37, test_diffuse.py:42-52) re-hosted: the histogram of sampled directions over a (cos theta, phi) grid against the
    integral of pdf() over each cell, cells with fewer than 5 expected samples pooled, significance 0.01 as in the reference.  `weight * pdf == eval`
    (test above) cannot see a sampler whose samples are not distributed like the pdf it reports; this can."""
    from scipy import stats
    P = Pair(mi, O, H, BSDF_DICTS[name])
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()) + int(os.environ.get('CHI2_SEED_OFFSET', '0')))
    n_samples, R, Cc, sub = 40000, 24, 48, 6
    for wi in ([0.0, 0.0, 1.0], [0.6, -0.3, 0.74]):
        wi = np.float32(wi) / np.linalg.norm(wi)
        hist = np.zeros((R, Cc))
        for _ in range(n_samples):
            s = rng.random(3)
            wo, pdf, w, eta, delta = P.sample(which, wi, float(s[0]), [float(s[1]), float(s[2])])
            if not (w > 0).any() or delta:         # invalid samples (e.g. a reflection below the horizon: weight 0) do not count, as in BSDFAdapter (chi2.py)
                continue
            z = min(max(float(wo[2]), -1.0), 1.0); phi = np.arctan2(float(wo[1]), float(wo[0])) % (2 * np.pi)
            hist[min(int((z + 1) / 2 * R), R - 1), min(int(phi / (2 * np.pi) * Cc), Cc - 1)] += 1
        # expected counts: midpoint rule on a sub x sub grid per cell (uniform measure dz dphi on the sphere)
        expected = np.zeros((R, Cc)); cell = (2.0 / R) * (2 * np.pi / Cc)
        for i in range(R):
            for j in range(Cc):
                acc = 0.0
                for a in range(sub):
                    for b in range(sub):
                        z = -1 + 2 * (i + (a + 0.5) / sub) / R; phi = 2 * np.pi * (j + (b + 0.5) / sub) / Cc
                        r = np.sqrt(max(0.0, 1 - z * z))
                        acc += P.eval_pdf(which, wi, [r * np.cos(phi), r * np.sin(phi), z])[1]
                expected[i, j] = acc / (sub * sub) * cell * n_samples
        assert abs(expected.sum() / n_samples - hist.sum() / n_samples) < 0.02          # the pdf's mass == the fraction of valid samples
        o, e = hist.ravel(), expected.ravel()
        order = np.argsort(e); o, e = o[order], e[order]
        pooled_o, pooled_e, chi2, dof = 0.0, 0.0, 0.0, 0
        for oo, ee in zip(o, e):
            if ee < 5:
                pooled_o += oo; pooled_e += ee
                continue
            chi2 += (oo - ee) ** 2 / ee; dof += 1
        if pooled_e > 0:
            chi2 += (pooled_o - pooled_e) ** 2 / max(pooled_e, 1e-9); dof += 1
        p = stats.chi2.sf(chi2, dof - 1)
        # 16 tests per implementation (8 models x 2 incident directions): Sidak-corrected level, like ChiSquareTest.run(significance_level, test_count)
        assert p > 1.0 - (1.0 - 0.01) ** (1.0 / 16.0), (name, which, wi, chi2, dof, p)
        print('chi2 %s %s wi.z=%.2f: p = %.3f' % (name, which, wi[2], p))
