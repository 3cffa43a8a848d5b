"""Environment-map emitter: the oracle's Hierarchical2D<Float, 0> and EnvironmentMapEmitter restatements against
the reference's own expectations -- src/core/tests/test_distr_2d.py:7-50 (Mathematica spot checks), :96-125 (forward/inverse identity),
src/emitters/tests/test_envmap.py:13-95 (chi^2-style density check, sampling-weight bounds) -- then the product's host build and
host-compiled device code against the oracle."""
import numpy as np
import pytest


def bilinear_to_square(v00, v10, v01, v11, x, y):
    """warp.h:497-513 restated in float64 (test-side expectation of test_distr_2d.py:37-48)"""
    def lti(v0, v1, s): return s * ((2 - s) * v0 + s * v1) / (v0 + v1) if abs(v0 - v1) > 1e-4 * (v0 + v1) else s
    lerp = lambda a, b, t: a + (b - a) * t
    r0, r1 = v00 + v10, v01 + v11
    c0, c1 = lerp(v00, v01, y), lerp(v10, v11, y)
    return (lti(c0, c1, x), lti(r0, r1, y)), lerp(c0, c1, x)


@pytest.mark.parametrize("normalize", [True, False])
def test_hier2d_reference_spot_checks(O, normalize):
    ref = np.float32([[1, 2, 5], [9, 7, 2]])
    intg = np.array([19, 16]) / 35
    d = O.Hier2D(ref, normalize=normalize)
    s = 35 / 8.0 if not normalize else 1
    def close(got, pos, pdf):
        return np.allclose(got[0][0], pos, atol=1e-6) and np.allclose(got[1][0], pdf, atol=1e-6 * max(1, s))
    assert close(d.sample([0, 0]), [0, 0], s * 8 / 35)
    assert close(d.sample([1, 1]), [1, 1], s * 16 / 35)
    assert close(d.sample([intg[0], 0]), [0.5, 0], s * 16 / 35)
    assert close(d.invert([0, 0]), [0, 0], s * 8 / 35)
    assert close(d.invert([1, 1]), [1, 1], s * 16 / 35)
    assert close(d.invert([0.5, 0]), [intg[0], 0], s * 16 / 35)
    (sx, sy), pdf = bilinear_to_square(1, 2, 9, 7, 0.4, 0.3)
    sx *= intg[0]; pdf *= 8 / 35 * s
    assert close(d.sample([sx, sy]), [0.2, 0.3], pdf) and close(d.invert([0.2, 0.3]), [sx, sy], pdf) and np.allclose(d.eval([0.2, 0.3]), pdf, atol=1e-6 * max(1, s))
    (sx, sy), pdf = bilinear_to_square(2, 5, 7, 2, 0.4, 0.3)
    sx = sx * intg[1] + intg[0]; pdf *= 8 / 35 * s
    assert close(d.sample([sx, sy]), [0.7, 0.3], pdf) and close(d.invert([0.7, 0.3]), [sx, sy], pdf) and np.allclose(d.eval([0.7, 0.3]), pdf, atol=1e-6 * max(1, s))


def test_hier2d_forward_inverse_and_density(O):
    rng = np.random.default_rng(0)
    for i in range(12):
        shape = rng.integers(2, 40, 2)
        values = (rng.random(shape) * 10).astype(np.float32)
        if i == 9: values = np.ones(shape, np.float32)
        if i == 10: values[rng.random(shape) < 0.7] = 0
        d = O.Hier2D(values)
        p_i = rng.random((2000, 2)).astype(np.float32)
        p_o, pdf = d.sample(p_i)
        assert np.allclose(pdf, d.eval(p_o), atol=1e-4 * max(1, pdf.max()))
        p_i2, pdf2 = d.invert(p_o)
        ok = pdf > 1e-3
        assert np.allclose(pdf[ok], pdf2[ok], atol=1e-4 * max(1, pdf.max())) and np.abs(p_i2[ok] - p_i[ok]).max() < 2e-3
        # density: histogram of warped samples vs integral of the normalised interpolant
        n = 400000
        po, _ = d.sample(rng.random((n, 2)).astype(np.float32))
        H, _, _ = np.histogram2d(po[:, 1], po[:, 0], bins=[4, 4], range=[[0, 1], [0, 1]])
        g = (np.stack(np.meshgrid((np.arange(64) + .5) / 64, (np.arange(64) + .5) / 64), -1).reshape(-1, 2)).astype(np.float32)
        dens = d.eval(g).reshape(64, 64).reshape(4, 16, 4, 16).mean(axis=(1, 3)) / 16
        assert np.abs(H / n - dens).max() < 0.01


def one_pixel_image():
    img = np.zeros((100, 10, 3), np.float32); img[40, 5] = 1
    return img


def test_envmap_sampling_weights_bounded(O):
    """test_envmap.py:45-95: envmap zero except one pixel; sample_direction weight == eval / pdf_direction, bounded in (0.018, 0.02)"""
    em = O.EnvMap(one_pixel_image())
    rng = np.random.default_rng(1)
    sample = rng.random((102400, 2)).astype(np.float32)
    d, dist, pdf, w = em.sample_direction([0, 0, 0], sample)
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-5) and np.all(dist == 2.0) and np.all(pdf > 0)
    w2 = em.eval(d) / em.pdf_direction(d)[:, None]
    rel = np.abs(w - w2) / np.abs(w2)
    assert (rel > 1e-3).mean() < 1e-4 and rel.max() < 5e-3          # (a handful of samples next to the zero texels round differently)
    assert w[:, 0].min() > 0.018 and w[:, 0].max() < 0.02


@pytest.mark.parametrize("img", ["pixel", "const_hi", "const_lo", "random"])
def test_envmap_density_matches_pdf(O, img):
    """test_envmap.py:13-42 (chi^2 test) as a histogram comparison over the sphere"""
    rng = np.random.default_rng(2)
    data = {"pixel": one_pixel_image(), "const_hi": np.ones((100, 100, 3), np.float32), "const_lo": np.ones((3, 2, 3), np.float32),
            "random": rng.random((17, 31, 3)).astype(np.float32) ** 4}[img]
    em = O.EnvMap(data)
    n = 600000
    d, _, pdf, _ = em.sample_direction([0, 0, 0], rng.random((n, 2)).astype(np.float32))
    nb = (8, 16)
    cos_t = d[:, 1]; phi = np.arctan2(d[:, 0], -d[:, 2])
    H, _, _ = np.histogram2d(cos_t, phi, bins=nb, range=[[-1, 1], [-np.pi, np.pi]])
    # integrate pdf_direction over the same (cos theta, phi) cells: cells have equal solid angle 4 pi / (8 * 16)
    m = 24
    ct = -1 + 2 * (np.arange(nb[0] * m) + .5) / (nb[0] * m); ph = -np.pi + 2 * np.pi * (np.arange(nb[1] * m) + .5) / (nb[1] * m)
    CT, PH = np.meshgrid(ct, ph, indexing="ij"); st = np.sqrt(1 - CT ** 2)
    dirs = np.stack([np.sin(PH) * st, CT, -np.cos(PH) * st], -1).reshape(-1, 3).astype(np.float32)
    p = em.pdf_direction(dirs).reshape(nb[0], m, nb[1], m).mean(axis=(1, 3)) * (4 * np.pi / (nb[0] * nb[1]))
    assert abs(p.sum() - 1) < 0.02
    assert np.abs(H / n - p).max() < 0.004 + 0.03 * p.max()


def test_envmap_eval_layout_and_transform(O):
    """lat-long convention (envmap.cpp:436-459): +Y is the top row, -Z the image centre column... and to_world rotates it"""
    H, W = 16, 32
    img = np.zeros((H, W, 3), np.float32)
    img[:, :, 0] = np.linspace(0, 1, H)[:, None]                      # R encodes the row (theta), align-corners
    img[:, :, 1] = ((np.arange(W) + .5) / W)[None, :]                 # G encodes the column centre (phi / 2 pi)
    em = O.EnvMap(img, scale=2.0)
    up = em.eval([[0, 1, 0]])[0]; down = em.eval([[0, -1, 0]])[0]
    assert abs(up[0] - 0) < 1e-6 and abs(down[0] - 2.0) < 1e-5
    for u in (0.1, 0.37, 0.5, 0.82):
        phi = 2 * np.pi * u
        d = [np.sin(phi), 0, -np.cos(phi)]
        g = em.eval([d])[0, 1] / 2.0
        assert abs(g - u) < 1e-4                                       # bilinear in texel centres reproduces a linear ramp (away from the seam)
    # rotation about Y by 90 degrees: to_world maps local +X to world -Z ... the lookup uses the inverse
    c, s = 0.0, 1.0
    tw = [c, 0, -s, 0, 1, 0, s, 0, c, 0, 0, 0]; tl = [c, 0, s, 0, 1, 0, -s, 0, c, 0, 0, 0]      # column-major 3 x 4
    em2 = O.EnvMap(img, to_world=tw, to_local=tl)
    d_local = np.float32([[np.sin(1.0), 0.2, -np.cos(1.0)]]); d_local /= np.linalg.norm(d_local)
    M = np.float32(tw[:9]).reshape(3, 3).T
    d_world = d_local @ M.T
    assert np.allclose(em2.eval(d_world), O.EnvMap(img).eval(d_local), atol=1e-6)
    assert np.allclose(em2.pdf_direction(d_world), O.EnvMap(img).pdf_direction(d_local), rtol=1e-5)
    # mis_compensation lowers the density of dim regions (envmap.cpp:497-519) but keeps it normalised
    rng = np.random.default_rng(3); data = rng.random((12, 24, 3)).astype(np.float32); data[3, 5] = 50
    a, b = O.EnvMap(data), O.EnvMap(data, mis_compensation=True)
    dirs = rng.normal(size=(2000, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    pa, pb = a.pdf_direction(dirs), b.pdf_direction(dirs)
    assert (pb == 0).sum() > 100 and pb.max() > pa.max() and np.array_equal(a.eval(dirs), b.eval(dirs))
