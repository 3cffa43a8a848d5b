"""BASELINE config 1 ("Cornell box, scalar_rgb path integrator, 64spp on CPU"): the oracle's restatement of the reference's
scalar-variant driver -- spiral block order, Morton pixel order, per-pixel reseeding, discretised reconstruction filter,
block borders (integrator.cpp:190-274,398-446; spiral.cpp:27-73; imageblock.cpp:283-375)."""
import ctypes as C
import os

import numpy as np
import pytest


def test_morton_and_spiral(O):
    L = O.lib(); out = (C.c_uint32 * 2)()
    want = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (0, 2), (1, 2), (0, 3)]
    for m, w in enumerate(want):
        L.orc_morton_decode(m, out); assert (out[0], out[1]) == w
    L.orc_morton_decode(0xffff, out); assert (out[0], out[1]) == (255, 255)
    buf = (C.c_int32 * (5 * 64))()
    n = L.orc_spiral(96, 96, 32, 64, buf)              # 3 x 3 blocks: centre, right, down, left, left, up, up, right, right
    blocks = np.frombuffer(buf, np.int32).reshape(-1, 5)[:n]
    assert n == 9 and [tuple(b[:2] // 32) for b in blocks] == [(1, 1), (2, 1), (2, 2), (1, 2), (0, 2), (0, 1), (0, 0), (1, 0), (2, 0)]
    assert list(blocks[:, 4]) == list(range(9))
    n = L.orc_spiral(100, 70, 32, 64, buf)             # ragged: 4 x 3 blocks, every pixel covered exactly once
    blocks = np.frombuffer(buf, np.int32).reshape(-1, 5)[:n]
    cover = np.zeros((70, 100), int)
    for ox, oy, sx, sy, _ in blocks:
        cover[oy:oy + sy, ox:ox + sx] += 1
    assert n == 12 and (cover == 1).all()


def test_reference_spiral_known_answers(O):
    """src/render/tests/test_spiral.py: test02_small_film (a 15 x 12 film is ONE block) and test03_normal_film (318 x 322: 110 blocks, the first
    twelve spiral outwards from the centre block at (160, 160): right, down, left, left, up, up, right, right, right, down, down)"""
    L = O.lib(); buf = (C.c_int32 * (5 * 256))()
    n = L.orc_spiral(15, 12, 32, 256, buf)
    assert n == 1 and list(np.frombuffer(buf, np.int32)[:5]) == [0, 0, 15, 12, 0]
    n = L.orc_spiral(318, 322, 32, 256, buf)
    blocks = np.frombuffer(buf, np.int32).reshape(-1, 5)[:n]
    w = 32; c = np.array([160, 160])
    steps = [(0, 0), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (2, -1), (2, 0), (2, 1)]
    assert n == 110
    for b, (sx, sy) in zip(blocks, steps):
        assert tuple(b[:2]) == tuple(c + np.array([sx, sy]) * w) and tuple(b[2:4]) == (w, w)
    cover = np.zeros((322, 318), int)
    for ox, oy, sx, sy, _ in blocks:
        cover[oy:oy + sy, ox:ox + sx] += 1
    assert (cover == 1).all()


def test_config1_scalar_cornell(O):
    """64 spp Cornell box through the scalar driver: same estimator as the JIT-order driver (different sample streams),
    so the images agree statistically; every pixel receives exactly spp samples; block size follows the thread count."""
    res, spp = 48, 64
    sd, sensor = O.cornell_box(res, res)
    osc = O.OracleScene(sd)
    film_s, st, bs = osc.render_path_scalar(sensor, seed=0, spp=spp, max_depth=8, raw=True)
    assert bs == 32 and st.paths == res * res * spp
    _, _, bs8 = osc.render_path_scalar(sensor, seed=0, spp=1, max_depth=1, n_threads=8, raw=True)
    assert bs8 == 16                                    # (48/32)^2 = 4 blocks < 8 threads -> 16 x 16 blocks
    film_j, _ = osc.render_path(sensor, seed=0, spp=spp, max_depth=8, raw=True)
    img_s, img_j = O.develop(film_s), O.develop(film_j)
    assert np.isfinite(img_s).all() and img_s.min() >= 0
    assert abs(img_s.mean() / img_j.mean() - 1) < 0.02
    blur = lambda a: a.reshape(res // 8, 8, res // 8, 8, 3).mean(axis=(1, 3))
    assert np.abs(blur(img_s) - blur(img_j)).max() / blur(img_j).max() < 0.08
    # weights: eval_discretized() looks the filter up at the LEFT edge of 31 bins per radius (rfilter.h:70-79), which
    # over-estimates a decreasing filter by ~5 % per axis -> ~10 % more accumulated weight; develop() divides it out
    assert 1.07 < film_s[..., 3].sum() / film_j[..., 3].sum() < 1.13
    # determinism + seed sensitivity
    film_s2, _, _ = osc.render_path_scalar(sensor, seed=0, spp=spp, max_depth=8, raw=True)
    assert np.array_equal(film_s, film_s2)
    film_s3, _, _ = osc.render_path_scalar(sensor, seed=1, spp=spp, max_depth=8, raw=True)
    assert not np.array_equal(film_s, film_s3)


def test_scalar_directly_visible_emitter_kat(O):
    """src/integrators/tests/test_integrators.py:28-53 runs under `variants_all_rgb`, i.e. also scalar_rgb: pixel (124, 36)"""
    sd, sensor = O.cornell_box(256, 256, crop=(124, 36, 1, 1))
    img, _, _ = O.OracleScene(sd).render_path_scalar(sensor, spp=64, max_depth=1)
    assert np.allclose(img.reshape(3), [18.387, 13.9873, 6.75357], rtol=1e-5)


def test_multipass_oracle(O):
    """JIT multi-pass render (integrator.cpp:173-183,276-356): lane i keeps its pixel and its sampler stream across passes"""
    res = 16
    sd, sensor = O.cornell_box(res, res)
    osc = O.OracleScene(sd)
    one, _ = osc.render_path(sensor, seed=5, spp=2, max_depth=6, raw=True)
    same, _ = osc.render_path_passes(sensor, seed=5, spp=2, spp_per_pass=2, max_depth=6, raw=True)
    assert np.array_equal(one, same)                                   # a single pass is the plain render
    multi, st = osc.render_path_passes(sensor, seed=5, spp=8, spp_per_pass=2, max_depth=6, raw=True)
    assert st.paths == res * res * 8
    assert abs(multi[..., 3].sum() / one[..., 3].sum() - 4) < 1e-3      # four passes of the same wavefront size
    assert np.abs(multi - 4 * one).max() > 1e-2                          # ... with fresh random numbers
    # union of two lane bands of the per-pass wavefront == whole job
    n = res * res * 2
    a, _ = osc.render_path_passes(sensor, seed=5, spp=8, spp_per_pass=2, max_depth=6, lanes=(0, n // 2), raw=True)
    b, _ = osc.render_path_passes(sensor, seed=5, spp=8, spp_per_pass=2, max_depth=6, lanes=(n // 2, n), raw=True)
    assert np.abs(a + b - multi).max() <= 1e-5 * np.abs(multi).max()
    # same estimator as one big wavefront
    big, _ = osc.render_path(sensor, seed=5, spp=1024, max_depth=6)
    mp, _ = osc.render_path_passes(sensor, seed=5, spp=1024, spp_per_pass=16, max_depth=6)
    assert abs(mp.mean() / big.mean() - 1) < 0.03                        # (image-mean noise at this size: ~1 %)


def test_pass_layout_host():
    """har_render_pass_layout mirrors integrator.cpp:173-183,276-294 incl. the reference's C5 quirk"""
    import ctypes as C
    from mitsuba3_amd import _capi
    L = _capi.lib()
    sensor = _capi.HarSensor(); sensor.crop_width = 4096; sensor.crop_height = 4096
    h = C.c_void_p(); _capi.check(L.har_integrator_create(0, 8, 5, 0, C.byref(h)))
    a, b = C.c_uint32(), C.c_uint32()
    # 4096^2 x 1024 = 2^34 lanes: 1024 / ceil(2^34 / (2^32 - 1)) = 204 samples per pass, which does not divide 1024 -> the reference throws
    assert L.har_render_pass_layout(h, C.byref(sensor), 1024, C.byref(a), C.byref(b)) != 0
    assert b"multiple" in L.har_last_error()
    _capi.check(L.har_integrator_set_samples_per_pass(h, 128))
    _capi.check(L.har_render_pass_layout(h, C.byref(sensor), 1024, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (128, 8)
    _capi.check(L.har_integrator_set_samples_per_pass(h, 256))       # 2^32 lanes > 2^32 - 1: halved again
    _capi.check(L.har_render_pass_layout(h, C.byref(sensor), 1024, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (128, 8)
    _capi.check(L.har_integrator_set_samples_per_pass(h, 48))
    assert L.har_render_pass_layout(h, C.byref(sensor), 1024, C.byref(a), C.byref(b)) != 0      # integrator.cpp:177-179
    sensor.crop_width = sensor.crop_height = 512
    _capi.check(L.har_integrator_set_samples_per_pass(h, 0))
    _capi.check(L.har_render_pass_layout(h, C.byref(sensor), 256, C.byref(a), C.byref(b)))
    assert (a.value, b.value) == (256, 1)
    L.har_integrator_destroy(h)
    hp = C.c_void_p(); _capi.check(L.har_integrator_create(1, 6, 5, 0, C.byref(hp)))
    assert L.har_integrator_set_samples_per_pass(hp, 4) != 0         # AD integrators render one wavefront (common.py:358-363)
    L.har_integrator_destroy(hp)


def test_ztest_drivers_agree(O):
    """the reference's Z-test (test_renders.py:146-236) between the oracle's drivers: scalar (spiral / Morton / discretised filter -- box filter
    here so that pixels are independent), multi-pass and single-wavefront renders estimate the same image"""
    from tests import ztest
    res = 24
    sd, sensor = O.cornell_box(res, res, rfilter="box")
    osc = O.OracleScene(sd)
    ref_mean, ref_var, n_ref = ztest.oracle_reference(osc, sensor, spp_b=8, batches=96, max_depth=6)
    scalar, _, _ = osc.render_path_scalar(sensor, seed=77, spp=256, max_depth=6)
    ok, pmin, alpha = ztest.accept(scalar, 256, ref_mean, ref_var, n_ref)
    assert ok, (pmin, alpha)
    multi, _ = osc.render_path_passes(sensor, seed=78, spp=256, spp_per_pass=32, max_depth=6)
    ok, pmin, alpha = ztest.accept(multi, 256, ref_mean, ref_var, n_ref)
    assert ok, (pmin, alpha)
    # the test has power: a 5 % brighter image is rejected
    ok, _, _ = ztest.accept(multi * 1.05, 256, ref_mean, ref_var, n_ref)
    assert not ok


# ------------------------------------------------------------------ the PRODUCT's scalar path (har_render_scalar, variant 'scalar_rgb')

def _rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def test_product_scalar_path_matches_the_oracle_scalar_driver(O):
    """config 1 in the product: har_render_scalar (Spiral, Morton order, per-pixel reseed, discretised filter, bordered blocks, put_block; the
    path code is the host compilation of the kernels' headers with scalar draw semantics) against the oracle's restatement of the same
    reference code -- same sample streams, so the films agree to rounding; no GPU involved"""
    import mitsuba3_amd as mi
    from mitsuba3_amd import core
    mi.set_variant("scalar_rgb")
    try:
        for res, spp, crop, rf, sb in ((48, 16, None, "gaussian", False), (40, 8, (7, 5, 21, 30), "gaussian", False), (32, 8, None, "box", False), (32, 4, None, "tent", False),
                                       (36, 4, None, "gaussian", True), (40, 4, (7, 5, 21, 30), "tent", True)):       # Film::sample_border: spiral over the enlarged film
            d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = res; f["height"] = res; f["rfilter"] = {"type": rf}; f["sample_border"] = sb
            if crop:
                f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = crop
            scene = mi.load_dict(d)
            osc, sensor = O.scene_from_product(scene)
            ref, st, bs = osc.render_path_scalar(sensor, seed=3, spp=spp, max_depth=8)
            img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 3, spp, threads=1)
            assert _rel_l2(img, ref) < 1e-5, (res, spp, crop, rf)
            img4 = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 3, spp, threads=8)       # other block size, other put_block order
            ref4, _, bs4 = osc.render_path_scalar(sensor, seed=3, spp=spp, max_depth=8, n_threads=8)
            assert _rel_l2(img4, ref4) < 1e-5 and (bs4 != bs or res * res <= 32 * 32 * 8)
        # mi.render goes through the same entry point under the scalar variant; the KAT of test_integrators.py:28-53 (runs under scalar_rgb too)
        d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = 256; f["height"] = 256
        f["crop_offset_x"], f["crop_offset_y"], f["crop_width"], f["crop_height"] = 124, 36, 1, 1
        d["integrator"] = {"type": "path", "max_depth": 1}
        img = mi.render(mi.load_dict(d), spp=64)
        assert np.allclose(np.asarray(img).reshape(3), [18.387, 13.9873, 6.75357], rtol=1e-5)
        # luminance / xyz films (hdrfilm.cpp:149-176): the develop step's colour transform of the same film (spectrum.h:402-442)
        for pf in ("luminance", "xyz"):
            d = mi.cornell_box(); f = d["sensor"]["film"]; f["width"] = 24; f["height"] = 24; f["pixel_format"] = pf
            scene = mi.load_dict(d)
            osc, sensor = O.scene_from_product(scene)
            ref, _, _ = osc.render_path_scalar(sensor, seed=3, spp=4, max_depth=8)
            M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
            want = (ref @ M[1])[..., None] if pf == "luminance" else ref @ M.T
            img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 3, 4, threads=1)
            assert img.shape == want.shape and _rel_l2(img, want) < 1e-5, pf
        # materials (conditional emitter draws where the BSDF has no smooth lobe: dielectric / conductor)
        d = mi.instanced_spheres_scene(width=24, height=24, spp=8, grid=2, n_u=8, n_v=4, flatten=True, materials=True)
        scene = mi.load_dict(d)
        osc, sensor = O.scene_from_product(scene)
        ref, _, _ = osc.render_path_scalar(sensor, seed=1, spp=8, max_depth=8)
        img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 1, 8, threads=1)
        assert _rel_l2(img, ref) < 1e-4
        # the plugins round 4 added, under the scalar variant too: point / spot / directional emitters behind an orthographic camera (tests/test_golden_cpu.py: round4_scene)
        from tests.test_golden_cpu import round4_scene
        scene = mi.load_dict(round4_scene(mi))
        osc, sensor = O.scene_from_product(scene)
        ref, _, _ = osc.render_path_scalar(sensor, seed=2, spp=8, max_depth=8)
        img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], 2, 8, threads=1)
        assert np.abs(ref).max() > 0 and _rel_l2(img, ref) < 1e-5
    finally:
        mi.set_variant("hip_ad_rgb")


def test_hip_variant_never_falls_back_to_the_scalar_path():
    """the scalar entry point is an explicit choice: under `hip_ad_rgb` a render without a GPU is an error, not a CPU render"""
    import torch
    import mitsuba3_amd as mi
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    mi.set_variant("hip_ad_rgb")
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 8; d["sensor"]["film"]["height"] = 8
    with pytest.raises(Exception) as e:
        mi.render(mi.load_dict(d), spp=1)
    assert "HIP device" in str(e.value) or "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("extended", [False, True])
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("HAR_SCALAR_SEEDS", "12")))))
def test_product_scalar_path_on_random_scenes(O, seed, extended):
    """the randomised scenes of tests/test_gpu_fuzz_parity.py (every shape / BSDF / emitter / sensor / filter of the variant in random combinations) through config 1's
    driver: har_render_scalar -- the HOST compilation of the kernels' path code with scalar draw semantics -- against the oracle's scalar driver, same streams; no GPU"""
    import mitsuba3_amd as mi
    from mitsuba3_amd import core
    from tests.test_gpu_fuzz_parity import random_scene
    mi.set_variant("scalar_rgb")
    try:
        d, cfg = random_scene(mi, int(os.environ.get("HAR_FUZZ_SEED0", "0")) + 300 + seed)
        if extended:
            from tests.test_gpu_fuzz_parity import extend_scene
            d = extend_scene(mi, d, seed)
        d["integrator"] = {"type": "path", "max_depth": cfg["max_depth"], "rr_depth": cfg["rr_depth"]}
        scene = mi.load_dict(d)
        osc, sensor = O.scene_from_product(scene)
        spp = min(cfg["spp"], 8)
        nt = 1 + 7 * (seed % 2)                          # (the block size follows the thread count, integrator.cpp:200-216: both sides get the same)
        ref, st, _ = osc.render_path_scalar(sensor, seed=seed, spp=spp, max_depth=cfg["max_depth"], rr_depth=cfg["rr_depth"], n_threads=nt)
        img = core._render_scalar(scene, scene.integrator(), scene.sensors()[0], seed, spp, threads=nt)
        assert np.isfinite(img).all()
        err = float(np.linalg.norm(img.astype(np.float64) - ref.astype(np.float64))); scale = float(np.linalg.norm(ref.astype(np.float64)))
        assert err <= 1e-5 * scale + 1e-9, (err / max(scale, 1e-30), cfg)
    finally:
        mi.set_variant("hip_ad_rgb")
