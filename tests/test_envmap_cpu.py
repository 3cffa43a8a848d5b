"""Environment-map emitter: the oracle's Hierarchical2D<Float, 0> and EnvironmentMapEmitter restatements against
the reference's own expectations -- src/core/tests/test_distr_2d.py:7-50 (Mathematica spot checks), :96-125 (forward/inverse identity),
src/emitters/tests/test_envmap.py:13-95 (chi^2-style density check, sampling-weight bounds) -- then the product's host build and
host-compiled device code against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p
    L.hh_scene_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.hh_scene_destroy.argtypes = [C.c_void_p]
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    L.hh_envmap_storage.argtypes = [C.c_void_p, O.c_u32p, O.c_u32p, O.c_f32p]
    L.hh_envmap_eval.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p]
    L.hh_envmap_sample_direction.argtypes = [C.c_void_p, C.c_uint32] + [O.c_f32p] * 6
    return L


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


# ---------------------------------------------------------------- product: host lowering + host-compiled device code vs the oracle

def _env_scene(mi, res=24, to_world=None, mis=False, with_area_light=True, seed=5):
    rng = np.random.default_rng(seed)
    env = (rng.random((12, 24, 3)).astype(np.float32) ** 3) * 2; env[2, 7] = [40, 30, 20]      # a "sun"
    d = mi.cornell_box() if with_area_light else {"type": "scene", "integrator": {"type": "path", "max_depth": 6},
                                                   "sensor": mi.cornell_box()["sensor"], "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.7, 0.6, 0.5]}},
                                                   "box": {"type": "cube", "to_world": mi.ScalarTransform4f().scale(0.4), "bsdf": {"type": "ref", "id": "white"}},
                                                   "floor": {"type": "rectangle", "to_world": mi.ScalarTransform4f().translate([0, -0.4, 0]).rotate([1, 0, 0], -90).scale(2), "bsdf": {"type": "ref", "id": "white"}}}
    if with_area_light:
        for k in ("ceiling", "back"):          # open the box so that the environment is visible
            d.pop(k, None)
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["env"] = {"type": "envmap", "bitmap": mi.Bitmap(env), "scale": 0.5, "mis_compensation": mis}
    if to_world is not None:
        d["env"]["to_world"] = to_world
    return mi.load_dict(d), env


def test_product_envmap_host_build_and_device_code(mi, O, H):
    from tests.test_cpu_host import oracle_scene_from, _harness_scene
    import ctypes as C
    T = mi.ScalarTransform4f
    for tw, mis in ((None, False), (T().rotate([0, 1, 0], 40).rotate([1, 0, 0], 15), True)):
        scene, env = _env_scene(mi, to_world=tw, mis=mis)
        h = _harness_scene(H, scene)
        e = scene.emitters[[i for i, x in enumerate(scene.emitters) if x["type"] == 2][0]]
        ref = O.EnvMap(env, scale=0.5, mis_compensation=mis, to_world=e["to_world"], to_local=e["to_local"])
        # storage of the hierarchical warp: identical layout and values
        info = (C.c_uint32 * 4)(); H.hh_envmap_storage(h, info, None, None)
        table = np.zeros((info[2], 2), np.uint32); warp = np.zeros(info[3], np.float32)
        H.hh_envmap_storage(h, info, O.up(table), O.fp(warp))
        # compare with an oracle Hier2D built from the same luminance grid via sampling behaviour (storage is private to each side)
        rng = np.random.default_rng(0)
        dirs = rng.normal(size=(20000, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rgb = np.zeros_like(dirs); pdf = np.zeros(len(dirs), np.float32)
        H.hh_envmap_eval(h, len(dirs), O.fp(dirs), O.fp(rgb), O.fp(pdf))
        assert np.allclose(rgb, ref.eval(dirs), rtol=2e-5, atol=1e-6)
        assert np.allclose(pdf, ref.pdf_direction(dirs), rtol=2e-4, atol=1e-7)
        s = rng.random((20000, 2)).astype(np.float32); p = np.zeros((20000, 3), np.float32)
        d = np.zeros((20000, 3), np.float32); dist = np.zeros(20000, np.float32); pdf = np.zeros(20000, np.float32); w = np.zeros((20000, 3), np.float32)
        H.hh_envmap_sample_direction(h, 20000, O.fp(p), O.fp(s), O.fp(d), O.fp(dist), O.fp(pdf), O.fp(w))
        ref.set_bsphere([0, 0, 0], 1.0)
        rd, rdist, rpdf, rw = ref.sample_direction([0, 0, 0], s)
        assert np.allclose(d, rd, atol=2e-6) and np.allclose(pdf, rpdf, rtol=1e-4) and np.allclose(w, rw, rtol=2e-4, atol=1e-6)
        assert int(info[0]) == 24 and int(info[1]) == 12 and warp.size >= 25 * 12
        H.hh_scene_destroy(h)


@pytest.mark.parametrize("mode,area", [(0, True), (1, True), (0, False)])
def test_product_envmap_shading_matches_oracle(mi, O, H, mode, area):
    """host-compiled shade_lane (envmap variant) vs the oracle's path / prb samplers on an environment-lit scene"""
    from tests.test_cpu_host import oracle_scene_from, _harness_scene, rel_l2
    import ctypes as C
    scene, _ = _env_scene(mi, res=24, to_world=mi.ScalarTransform4f().rotate([0, 1, 0], 70), with_area_light=area)
    osc, sensor = oracle_scene_from(O, scene)
    h = _harness_scene(H, scene)
    film = np.zeros((24, 24, 4), np.float32)
    md = 6
    assert H.hh_render(h, C.byref(sensor), mode, 3, 8, md, 5, 0, 0, O.fp(film)) == 0
    ref, _ = (osc.render_path if mode == 0 else osc.render_prb)(sensor, seed=3, spp=8, max_depth=md, raw=True, threads=2)
    assert O.develop(ref).mean() > 0.05
    assert rel_l2(O.develop(film), O.develop(ref)) < 2e-5
    H.hh_scene_destroy(h)


def test_bitmap_read_roundtrip_and_envmap_from_file(mi, O, tmp_path):
    """Bitmap(filename): what write() produces (uncompressed EXR, PFM) plus ZIP / ZIPS / HALF OpenEXR files written by the test"""
    import os, struct, zlib
    rng = np.random.default_rng(7)
    for c in (1, 3, 4):
        img = rng.random((9, 13, c)).astype(np.float32) * 10
        p = os.path.join(tmp_path, "a%d.exr" % c); mi.Bitmap(img).write(p)
        assert np.array_equal(mi.Bitmap(p).data, img)
        if c != 4:
            p = os.path.join(tmp_path, "a%d.pfm" % c); mi.Bitmap(img).write(p)
            assert np.array_equal(mi.Bitmap(p).data, img)
    # big-endian PFM
    img = rng.random((4, 5, 3)).astype(np.float32)
    p = os.path.join(tmp_path, "be.pfm")
    with open(p, "wb") as f: f.write(b"PF\n5 4\n1.0\n" + img[::-1].astype(">f4").tobytes())
    assert np.array_equal(mi.Bitmap(p).data, img)

    def write_exr(path, img, comp, half, window=(0, 0)):
        H_, W_, Cn = img.shape
        names = {1: ["Y"], 3: ["B", "G", "R"], 4: ["A", "B", "G", "R"]}[Cn]; src = {1: [0], 3: [2, 1, 0], 4: [3, 2, 1, 0]}[Cn]
        def attr(n, t, d): return n.encode() + b"\0" + t.encode() + b"\0" + struct.pack("<i", len(d)) + d
        ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if half else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
        x0, y0 = window; win = struct.pack("<iiii", x0, y0, x0 + W_ - 1, y0 + H_ - 1)
        hdr = struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", ch) + attr("compression", "compression", bytes([comp])) + attr("dataWindow", "box2i", win) + \
            attr("displayWindow", "box2i", win) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + \
            attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
        lpc = 16 if comp == 3 else 1
        chunks = []
        for yb in range(0, H_, lpc):
            raw = b""
            for y in range(yb, min(yb + lpc, H_)):
                for k in src: raw += img[y, :, k].astype("<f2" if half else "<f4").tobytes()
            data = raw
            if comp in (2, 3):
                a = np.frombuffer(raw, np.uint8); t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
                dlt = t.copy(); dlt[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
                z = zlib.compress(dlt.astype(np.uint8).tobytes())
                data = z if len(z) < len(raw) else raw
            chunks.append(struct.pack("<ii", y0 + yb, len(data)) + data)
        off = len(hdr) + 8 * len(chunks); table = b""
        for c_ in chunks: table += struct.pack("<Q", off); off += len(c_)
        with open(path, "wb") as f: f.write(hdr + table + b"".join(chunks))

    smooth = np.stack([np.outer(np.linspace(0, 1, 37), np.linspace(1, 2, 21))] * 3, -1).astype(np.float32) * np.float32([1, 0.5, 0.25])
    for comp in (0, 2, 3):
        for half in (False, True):
            p = os.path.join(tmp_path, "z.exr"); write_exr(p, smooth, comp, half, window=(3, -2))
            got = mi.Bitmap(p).data
            assert np.array_equal(got, smooth.astype(np.float16).astype(np.float32) if half else smooth)
    write_exr(os.path.join(tmp_path, "y.exr"), smooth[:, :, :1], 3, False)
    assert np.array_equal(mi.Bitmap(os.path.join(tmp_path, "y.exr")).data, smooth[:, :, :1])
    with pytest.raises(RuntimeError, match="not found"): mi.Bitmap(os.path.join(tmp_path, "missing.exr"))
    with open(os.path.join(tmp_path, "bad.exr"), "wb") as f: f.write(b"not an image at all")
    with pytest.raises(RuntimeError, match="unknown file format"): mi.Bitmap(os.path.join(tmp_path, "bad.exr"))
    # envmap from a file == envmap from the bitmap (test_envmap.py:98-131); 1-channel images are replicated
    one = np.zeros((100, 10, 1), np.float32); one[40, 5] = 1
    p = os.path.join(tmp_path, "out.exr"); mi.Bitmap(one).write(p)
    a = mi.load_dict({"type": "envmap", "filename": p}); b = mi.load_dict({"type": "envmap", "bitmap": mi.Bitmap(one)})
    assert a.data.shape == (100, 10, 3) and np.array_equal(a.data, b.data)
    with pytest.raises(RuntimeError, match="both"): mi.load_dict({"type": "envmap", "filename": p, "bitmap": mi.Bitmap(one)})
    d = mi.cornell_box(); d["e1"] = {"type": "envmap", "bitmap": mi.Bitmap(one)}; d["e2"] = {"type": "constant"}
    with pytest.raises(RuntimeError, match="Only one environment emitter"): mi.load_dict(d)


def _eval_uv(em, u, v):
    """radiance at lat-long coordinates (u, v) in the convention of the reference's test helper (test_envmap.py:176-197)"""
    u = np.atleast_1d(np.asarray(u, np.float64)); v = np.atleast_1d(np.asarray(v, np.float64))
    phi, theta = u * 2 * np.pi, v * np.pi
    st, ct = np.sin(theta), np.cos(theta)
    d = np.stack([np.sin(phi) * st, ct, -np.cos(phi) * st], axis=1).astype(np.float32)
    return em.eval(d)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_reference_envmap_pixel_shift_seam_and_poles(mi, O, H, which):
    """src/emitters/tests/test_envmap.py: test06_pixel_shift (a ramp reads back the texel-centre coordinate u W - 1/2 along phi and the
    align-corners coordinate v (H - 1) along theta), test07_phi_seam_continuity (a periodic cosine is continuous across u = 0) and
    test08_theta_pole_clamp (the poles clamp, they do not wrap) -- with the reference's non-hardware-texture tolerances; the oracle's EnvMap and the
    product's host-built tables + host-compiled device lookup"""
    if which == "oracle":
        make = lambda img: O.EnvMap(np.asarray(img, np.float32))
    else:
        class Prod:
            def __init__(self, img):
                d = {"type": "scene", "integrator": {"type": "path", "max_depth": 2}, "sensor": mi.cornell_box()["sensor"],
                     "env": {"type": "envmap", "bitmap": mi.Bitmap(np.asarray(img, np.float32))}}
                self.scene = mi.load_dict(d); desc = self.scene.desc(); err = C.create_string_buffer(256)
                self.h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256)); assert self.h, err.value
            def eval(self, dirs):
                dirs = np.ascontiguousarray(dirs, np.float32); n = dirs.shape[0]; out = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
                H.hh_envmap_eval(self.h, n, O.fp(dirs), O.fp(out), O.fp(pdf)); return out
        make = Prod
    W, Hh = 16, 8
    img = np.broadcast_to(np.arange(W, dtype=np.float32)[None, :, None], (Hh, W, 3)).copy()
    t = np.linspace(1.0 / W, 1.0 - 1.0 / W, 50)
    assert np.abs(_eval_uv(make(img), t, np.full_like(t, 0.5))[:, 0] - (t * W - 0.5)).max() < 1e-4
    W, Hh = 8, 16
    img = np.broadcast_to(np.arange(Hh, dtype=np.float32)[:, None, None], (Hh, W, 3)).copy()
    t = np.linspace(1.0 / Hh, 1.0 - 1.0 / Hh, 50)
    assert np.abs(_eval_uv(make(img), np.zeros_like(t), t)[:, 0] - t * (Hh - 1)).max() < 1e-4
    W, Hh = 64, 8
    col = np.cos(2 * np.pi * np.arange(W) / W).astype(np.float32)
    img = np.broadcast_to(col[None, :, None], (Hh, W, 3)).copy()
    u = np.linspace(-0.1, 0.1, 201); uw = u - np.floor(u)
    got = _eval_uv(make(img), u, np.full_like(u, 0.5))[:, 0]
    assert np.abs(got - np.cos(2 * np.pi * (uw * W - 0.5) / W)).max() < 5e-3
    W, Hh = 8, 16
    img = np.zeros((Hh, W, 3), np.float32); img[0] = 10.0; img[Hh - 1] = 20.0
    em = make(img)
    assert abs(_eval_uv(em, 0.0, 0.0)[0, 0] - 10.0) < 1e-3 and abs(_eval_uv(em, 0.0, 1.0)[0, 0] - 20.0) < 1e-3


def test_envmap_scale_and_to_world_are_updatable_parameters(mi, O):
    """EnvironmentMapEmitter::traverse registers `scale` and `to_world` (envmap.cpp:204-208): written + params.update(), the scene equals a freshly loaded one with those
    values (oracle render of the host mirrors, bit for bit)"""
    import torch
    T = mi.ScalarTransform4f
    rng = np.random.default_rng(3)
    bm = mi.Bitmap(rng.uniform(0.1, 1.5, (6, 12, 3)).astype(np.float32))
    def make(scale, tw):
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = 20; d["sensor"]["film"]["height"] = 20
        d.pop("light"); d.pop("ceiling")
        d["env"] = {"type": "envmap", "bitmap": bm, "scale": scale, "to_world": tw}
        return d
    tw0 = T().rotate([0, 1, 0], 30.0); tw1 = T().rotate([0, 1, 0], 140.0).rotate([1, 0, 0], 25.0)
    scene = mi.load_dict(make(1.0, tw0))
    params = mi.traverse(scene)
    assert float(params["env.scale"]) == 1.0 and tuple(params["env.to_world"].shape) == (4, 4)
    params["env.scale"] = torch.tensor([2.5]); params["env.to_world"] = torch.as_tensor(np.asarray(tw1.matrix, np.float32)); params.update()
    m = np.asarray(tw1.matrix, np.float64)
    want = mi.load_dict(make(2.5, T(np.concatenate([m.ravel(), np.linalg.inv(m).T.ravel()]).astype(np.float32))))
    o1, s1 = O.scene_from_product(scene); o2, s2 = O.scene_from_product(want)
    a, st1 = o1.render_path(s1, seed=2, spp=4, max_depth=4, threads=1); b, st2 = o2.render_path(s2, seed=2, spp=4, max_depth=4, threads=1)
    assert st1.vertices == st2.vertices and np.allclose(a, b, rtol=2e-6, atol=1e-7) and a.mean() > 0
    base, _ = O.scene_from_product(mi.load_dict(make(1.0, tw0)))[0].render_path(s1, seed=2, spp=4, max_depth=4, threads=1)
    assert np.linalg.norm(a - base) > 0.1 * np.linalg.norm(base)          # the update did change the picture


def test_envmap_data_is_a_parameter_in_the_references_layout(mi, O):
    """EnvironmentMapEmitter::traverse registers `data` = the texel tensor with a one-column halo on each side (H x (W + 2) x 3, real column x at x + 1: envmap.cpp:146-188);
    written + params.update(), the real columns are the new texels (the halo is refreshed from them, whatever the caller put there: :226-246) and the scene equals a freshly
    loaded one with that bitmap"""
    import torch
    rng = np.random.default_rng(5)
    a0 = rng.uniform(0.1, 1.5, (6, 10, 3)).astype(np.float32); a1 = rng.uniform(0.1, 2.5, (6, 10, 3)).astype(np.float32)
    def make(a):
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = 20; d["sensor"]["film"]["height"] = 20
        d.pop("light"); d.pop("ceiling")
        d["env"] = {"type": "envmap", "bitmap": mi.Bitmap(a)}
        return d
    scene = mi.load_dict(make(a0))
    params = mi.traverse(scene)
    t = params["env.data"].cpu().numpy()
    assert t.shape == (6, 12, 3) and np.array_equal(t[:, 1:-1], a0) and np.array_equal(t[:, 0], a0[:, -1]) and np.array_equal(t[:, -1], a0[:, 0])
    new = np.concatenate([np.zeros((6, 1, 3), np.float32), a1, np.full((6, 1, 3), 7.0, np.float32)], axis=1)       # a stale halo: ignored
    params["env.data"] = torch.as_tensor(new); params.update()
    o1, s1 = O.scene_from_product(scene); o2, s2 = O.scene_from_product(mi.load_dict(make(a1)))
    a, st1 = o1.render_path(s1, seed=2, spp=4, max_depth=4, threads=1); b, st2 = o2.render_path(s2, seed=2, spp=4, max_depth=4, threads=1)
    assert st1.vertices == st2.vertices and np.array_equal(a, b)
    t = mi.traverse(scene)["env.data"].cpu().numpy()
    assert np.array_equal(t[:, 1:-1], a1) and np.array_equal(t[:, 0], a1[:, -1])
    with pytest.raises(RuntimeError, match="channels"):
        params["env.data"] = torch.zeros((6, 12, 4)); params.update()
