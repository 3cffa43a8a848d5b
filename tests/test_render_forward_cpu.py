"""RBIntegrator.render_forward (src/python/python/ad/integrators/common.py:497-623) in the oracle: the forward-mode derivative image is pinned
two independent ways before the GPU tests compare the product with it --
  * finite differences of the primal render (the check of src/render/tests/test_ad.py:6-134 `test01_bsdf_reflectance_forward`): with Russian
    roulette off, a fixed sample stream makes the image a polynomial of the albedos and linear in the emitter radiance;
  * the adjoint identity <J t, g> = <t, J^T g> against render_backward, which has its own finite-difference and golden pins."""
import numpy as np
import pytest

from oracle import oracle as O


def _scene(res=20, tex_res=6):
    rng = np.random.default_rng(0)
    tex = rng.uniform(0.3, 0.8, (tex_res, tex_res, 3)).astype(np.float32)
    sd, sensor = O.cornell_box(res, res, white_texture=tex)
    return sd, sensor, tex


def test_forward_image_matches_finite_differences():
    res, spp, md, rr = 20, 16, 5, 100            # rr_depth > max_depth: no Russian roulette, the sample stream does not depend on the parameters
    sd, sensor, tex = _scene(res)
    osc = O.OracleScene(sd)
    rng = np.random.default_rng(1)
    t_tex = rng.uniform(-1, 1, tex.shape).astype(np.float32)
    nb = len(sd.bsdfs)
    t_refl = rng.uniform(-1, 1, (nb, 3)).astype(np.float32)
    t_emit = rng.uniform(-1, 1, (len(sd.emitters), 3)).astype(np.float32)
    fwd = osc.render_prb_forward(sensor, t_refl, [t_tex], t_emit, seed=4, spp=spp, max_depth=md, rr_depth=rr)
    eps = 2e-3
    imgs = []
    base_refl = [np.array(b[2], np.float32) for b in sd.bsdfs]
    base_emit = [np.array(e["radiance"], np.float32) for e in sd.emitters]
    for sgn in (+1, -1):
        osc.set_texture(0, tex + sgn * eps * t_tex)
        for i in range(nb):
            osc.set_reflectance(i, base_refl[i] + sgn * eps * t_refl[i])
        for i in range(len(sd.emitters)):
            osc.set_emitter_radiance(i, base_emit[i] + sgn * eps * t_emit[i])
        img, _ = osc.render_prb(sensor, seed=4, spp=spp, max_depth=md, rr_depth=rr)
        imgs.append(img.astype(np.float64))
    fd = (imgs[0] - imgs[1]) / (2 * eps)
    err = np.linalg.norm(fwd - fd) / np.linalg.norm(fd)
    assert err < 2e-3, err                        # float32 renders differenced at eps = 2e-3: ~1e-4 relative noise + O(eps^2) truncation


def test_forward_is_the_transpose_of_backward():
    res, spp, md = 20, 8, 6
    sd, sensor, tex = _scene(res)
    osc = O.OracleScene(sd)
    rng = np.random.default_rng(2)
    t_tex = rng.uniform(-1, 1, tex.shape).astype(np.float32)
    t_refl = rng.uniform(-1, 1, (len(sd.bsdfs), 3)).astype(np.float32)
    t_emit = rng.uniform(-1, 1, (len(sd.emitters), 3)).astype(np.float32)
    g = rng.uniform(-1, 1, (res, res, 3)).astype(np.float32)
    fwd = osc.render_prb_forward(sensor, t_refl, [t_tex], t_emit, seed=9, spp=spp, max_depth=md)     # default rr_depth: Russian roulette on
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, g, seed=9, spp=spp, max_depth=md)
    lhs = float((fwd.astype(np.float64) * g).sum())
    rhs = float((g_refl.astype(np.float64) * t_refl).sum() + (g_tex[0].astype(np.float64) * t_tex).sum() + (np.asarray(g_emit, np.float64) * t_emit).sum())
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs), 1e-6), (lhs, rhs)


def test_forward_is_linear_in_the_tangents():
    res, spp, md = 16, 4, 5
    sd, sensor, tex = _scene(res)
    osc = O.OracleScene(sd)
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, tex.shape).astype(np.float32); b = rng.uniform(-1, 1, tex.shape).astype(np.float32)
    z = np.zeros((len(sd.bsdfs), 3), np.float32)
    fa = osc.render_prb_forward(sensor, z, [a], None, seed=1, spp=spp, max_depth=md)
    fb = osc.render_prb_forward(sensor, z, [b], None, seed=1, spp=spp, max_depth=md)
    fab = osc.render_prb_forward(sensor, z, [2 * a - 3 * b], None, seed=1, spp=spp, max_depth=md)
    assert np.linalg.norm(fab - (2 * fa - 3 * fb)) / np.linalg.norm(fab) < 1e-5
    assert np.abs(osc.render_prb_forward(sensor, z, [0 * a], None, seed=1, spp=spp, max_depth=md)).max() == 0.0
