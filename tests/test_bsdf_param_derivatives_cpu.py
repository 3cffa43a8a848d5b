"""d value / d {alpha_u, alpha_v, eta, k} of the rough BSDF models (har_bsdf.h: bsdf_eval_extra_one, hand-derived) against central differences of
the product's own eval() in its parameters, on the host build of the headers.  roughconductor.cpp:429-520, roughplastic.cpp:296-336,
microfacet.h:185-207,341-365, fresnel.h:93-116."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    lib = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    lib.hh_bsdf_eval_extra.argtypes = [C.c_int] * 3 + [C.c_float] * 3 + [C.POINTER(C.c_float)] * 8
    return lib


def _ev(H, type_, ggx, au, av, ec, kc, s0, s1, wi, wo):
    f3 = lambda a: np.ascontiguousarray(a, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    v = np.zeros(3, np.float32); o = np.zeros(12, np.float32)
    a = [f3(x) for x in (ec, kc, s0, s1, wi, wo)]
    H.hh_bsdf_eval_extra(type_, ggx, 1, au, av, 1.5, *[fp(x) for x in a], fp(v), fp(o))
    return v.astype(np.float64), o.reshape(4, 3).astype(np.float64)


@pytest.mark.parametrize("type_", [2, 3], ids=["roughconductor", "roughplastic"])
@pytest.mark.parametrize("ggx", [0, 1], ids=["beckmann", "ggx"])
def test_parameter_derivatives_match_finite_differences(H, type_, ggx):
    rng = np.random.default_rng(7 + type_ + 10 * ggx)
    errs = []
    for _ in range(300):
        au = float(rng.uniform(0.08, 0.6)); av = au if type_ == 3 else float(rng.uniform(0.08, 0.6))
        ec = rng.uniform(0.2, 2.0, 3); kc = rng.uniform(1.0, 4.0, 3); s0 = rng.uniform(0.2, 1, 3); s1 = rng.uniform(0.2, 1, 3)

        def direction():
            v = rng.normal(size=3); v[2] = abs(v[2]) + 0.3
            return v / np.linalg.norm(v)
        wi, wo = direction(), direction()
        args = dict(ec=ec, kc=kc, s0=s0, s1=s1, wi=wi, wo=wo)
        v, g = _ev(H, type_, ggx, au, av, **args)
        if np.abs(v).max() == 0:            # wi / wo on different sides of the half vector: the model evaluates to zero, and so do the derivatives
            assert np.abs(g).max() == 0
            continue
        h = 2e-3
        if type_ == 2:
            fd = np.zeros((4, 3))
            fd[0] = (_ev(H, 2, ggx, au * (1 + h), av, **args)[0] - _ev(H, 2, ggx, au * (1 - h), av, **args)[0]) / (2 * h * au)
            fd[1] = (_ev(H, 2, ggx, au, av * (1 + h), **args)[0] - _ev(H, 2, ggx, au, av * (1 - h), **args)[0]) / (2 * h * av)
            for c in range(3):
                for row, key in ((2, "ec"), (3, "kc")):
                    p1 = dict(args); p2 = dict(args); a1 = args[key].copy(); a2 = args[key].copy(); a1[c] *= 1 + h; a2[c] *= 1 - h; p1[key] = a1; p2[key] = a2
                    fd[row, c] = (_ev(H, 2, ggx, au, av, **p1)[0][c] - _ev(H, 2, ggx, au, av, **p2)[0][c]) / (2 * h * args[key][c])
            spec_scale = np.abs(v).max()
        else:
            fd = np.zeros((4, 3))
            fd[0] = (_ev(H, 3, ggx, au * (1 + h), au * (1 + h), **args)[0] - _ev(H, 3, ggx, au * (1 - h), au * (1 - h), **args)[0]) / (2 * h * au)
            spec_scale = np.abs(v).max()
        # float32 central differences: noise ~ 1e-7 * |value| / (h * alpha); compare against that floor
        floor = 3e-4 * spec_scale / (h * min(au, av))
        errs.append(np.abs(g - fd).max() / (np.abs(fd).max() + floor))
    errs = np.array(errs)
    assert np.median(errs) < 2e-3 and (errs < 3e-2).mean() > 0.97, (np.median(errs), errs.max())


# ------------------------------------------------------------------ the oracle's render-level gradients of eta / k / specular_reflectance

def test_oracle_bsdf_param_gradients_vs_finite_differences():
    """orc_render_prb_backward_bsdf_params on a scene with rough conductor / rough plastic records: the BSDF *sampling* of these models does not
    depend on eta, k or the specular colour (only the microfacet distribution enters the pdf), so with Russian roulette off a fixed sample stream
    makes the image a smooth function of them and central differences of the primal render pin the whole gradient plumbing (throughput, MIS
    weight, replayed radiance).  `alpha` moves the sampled directions (detached in PRB, prb.py:220-223), so its render-level check is the
    statistical one below; its derivative arithmetic is pinned by the double-precision differences inside the oracle and the unit test above."""
    import mitsuba3_amd as mi
    from oracle import oracle as O
    mi.set_variant("hip_ad_rgb")
    res, spp, md = 24, 64, 4
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=2, n_u=8, n_v=4, flatten=True, materials=True)
    scene = mi.load_dict(d)                                  # host-side scene only
    osc, sensor = O.scene_from_product(scene)
    g = np.random.default_rng(0).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    gx, _refl = osc.render_prb_backward_bsdf_params(sensor, g, seed=5, spp=spp, max_depth=md, rr_depth=100)
    kinds = [b.kind for b in scene.bsdf_objs]
    rc = kinds.index("roughconductor"); rp = kinds.index("roughplastic")
    assert np.abs(gx[rc, 2:4]).max() > 0 and np.abs(gx[rp, 4]).max() > 0

    def loss_with(mut):
        sd = O.scene_from_product(scene)[0]                  # fresh oracle scene from the (mutated) product description
        img, _ = sd.render_prb(sensor, seed=5, spp=spp, max_depth=md, rr_depth=100)
        return float((img.astype(np.float64) * g).sum())

    gx, g_refl = gx, _refl
    for name, idx, group in (("eta_c", rc, 2), ("k_c", rc, 3)):
        b = scene.bsdf_objs[idx]
        base = np.array(getattr(b, name), np.float32).copy()
        for c in range(3):
            h = 2e-2 * max(abs(float(base[c])), 0.05)
            v = base.copy(); v[c] += h; setattr(b, name, v); lp = loss_with(None)
            v = base.copy(); v[c] -= h; setattr(b, name, v); lm = loss_with(None)
            setattr(b, name, base.copy())
            fd = (lp - lm) / (2 * h)
            assert abs(gx[idx, group, c] - fd) < 2e-2 * abs(fd) + 1e-4 * np.abs(gx[idx]).max(), (name, c, gx[idx, group, c], fd)
    # rough plastic picks its lobe with a probability s_mean / (d_mean + s_mean) of the colours' means (RoughPlastic::parameters_changed,
    # roughplastic.cpp:204-242 -- a detached scalar in the reference too), so the specular colour is differenced along directions that keep it:
    # e_c - e_c' (s_mean unchanged) and a common scale of both colour slots (ratio unchanged); together they span the three channels
    b = scene.bsdf_objs[rp]
    spec0, diff0 = np.array(b.value2, np.float32).copy(), np.array(b.value, np.float32).copy()
    def directional(d_spec, d_diff, h):
        b.value2 = (spec0 + h * d_spec).astype(np.float32); b.value = (diff0 + h * d_diff).astype(np.float32); lp = loss_with(None)
        b.value2 = (spec0 - h * d_spec).astype(np.float32); b.value = (diff0 - h * d_diff).astype(np.float32); lm = loss_with(None)
        b.value2, b.value = spec0.copy(), diff0.copy()
        return (lp - lm) / (2 * h)
    scale = np.abs(gx[rp, 4]).max() * np.abs(spec0).max()
    for c, c2 in ((0, 1), (1, 2)):
        dv = np.zeros(3, np.float32); dv[c] = 1; dv[c2] = -1
        fd = directional(dv, np.zeros(3, np.float32), 0.02)
        assert abs((gx[rp, 4, c] - gx[rp, 4, c2]) - fd) < 2e-2 * np.abs(gx[rp, 4]).max(), (c, c2, gx[rp, 4], fd)
    fd = directional(spec0, diff0, 0.02)
    an = float((gx[rp, 4] * spec0).sum() + (g_refl[rp] * diff0).sum())
    assert abs(an - fd) < 2e-2 * abs(fd) + 1e-3 * scale, (an, fd)
