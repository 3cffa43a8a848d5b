"""GPU parity on the HEADLINE workloads at (or near) BASELINE.json's own sizes, and the bench's launch pattern.

Round-1 verdict, row N1: north_star asks that forward images (1e-4) and PRB gradients (1e-3 relative L2) match "on the same scene
and seed" for the 1M-triangle scene and the BASELINE configs, not only on toy sizes.  Methodology of the reference:
src/python/python/ad/integrators/common.py:625-783 (render_backward), prb.py:68-339, src/integrators/tests/test_integrators.py:28-53.
The oracle (oracle/) is the checker; every product call goes through the C ABI (mitsuba3_amd/_capi.py -> libhip_ad_rgb.so).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def _scene_1m(mi, O, res, spp, flatten=False, textured=False, integrator=None, grid=10, n_u=100, n_v=50, materials=False):
    """product scene from the dict; ORACLE scene from the oracle's own lowering of the same description (its transform chain, mesh baking,
    instance matrices and sensor: `O.benchmark_spheres_scene`) -- not `scene_from_product`, which would hand the oracle the product's baked
    arrays and hide a defect of the product's host lowering (round-2 verdict, "shared lowering")"""
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=grid, n_u=n_u, n_v=n_v, flatten=flatten, textured=textured, materials=materials)
    if integrator:
        d["integrator"] = integrator
    scene = mi.load_dict(d)
    sd, sensor = O.benchmark_spheres_scene(res, res, grid=grid, n_u=n_u, n_v=n_v, flatten=flatten, textured=textured, materials=materials)
    return scene, O.OracleScene(sd), sensor


# oracle BSDF / emitter indices of the benchmark scenes (O.benchmark_spheres_scene: white, green, red; one emitter) by the product's parameter keys
_ORACLE_SLOT = {"green.reflectance.value": ("refl", 1), "red.reflectance.value": ("refl", 2), "white.reflectance.value": ("refl", 0),
                "light.emitter.radiance.value": ("emit", 0)}


def _check_prb_gradients(grads, g_refl, g_tex, g_emit, tol=1e-3):
    g = grads["white.reflectance.data"].cpu().numpy()
    assert g.shape == g_tex[0].shape and np.count_nonzero(g_tex[0]) > 0.5 * g.size
    assert rel_l2(g, g_tex[0]) < tol                                       # north_star PRB tolerance
    checked = 0
    for key, (kind, idx) in _ORACLE_SLOT.items():                          # constant albedos (green, red walls) and the emitter's radiance
        if key not in grads:
            continue
        ref = g_emit[idx] if kind == "emit" else g_refl[idx]
        assert np.abs(np.asarray(ref)).max() > 0
        assert rel_l2(grads[key].cpu().numpy().reshape(-1), np.asarray(ref).reshape(-1)) < tol, key
        checked += 1
    assert checked >= 3


# ------------------------------------------------------------------ the bench's launch pattern (BENCH_r01: GPU memory access fault)

def test_unsynchronised_profiled_frames_instanced1m(mi):
    """25 frames of the headline workload (instanced 1M triangles, 512 x 512 x 256 spp = one 2^26-lane wavefront) enqueued back-to-back through
    render_distributed with per-launch HIP events on and NO synchronisation in between -- exactly what `bench.py --steps 20 --warmup 5` does.
    The frames are the same job (same seed): the last one must equal the first, the device counters must agree, and the profile must cover
    every frame."""
    import torch
    res, spp, frames = 512, 256, 25
    scene = mi.load_dict(mi.instanced_spheres_scene(width=res, height=res, spp=spp))
    integ = scene.integrator()
    first = mi.render_distributed(scene, integ, seed=0, spp=spp)
    torch.cuda.synchronize()
    st0 = integ.stats()
    assert st0["paths"] == res * res * spp
    integ.set_profiling(True)
    last = None
    for _ in range(frames):
        last = mi.render_distributed(scene, integ, seed=0, spp=spp)          # earlier images are dropped while their kernels may still be queued
    torch.cuda.synchronize()
    assert integ.stats() == st0
    t = integ.timing()
    assert int(t["frames"][0]) == frames and t["trace_closest"][1] == 9 and t["total"][0] > 0        # 8 bounces; the camera rays take two launches (packets, then the rays of the packets that gave up)
    integ.set_profiling(False)
    assert bool(torch.isfinite(last).all())
    assert rel_l2(last.cpu().numpy(), first.cpu().numpy()) < 1e-6             # float atomics of the film splat commute up to rounding


def test_unsynchronised_prb_steps_textured_instanced1m(mi):
    """the PRB leg of the bench: un-synchronised render_backward_distributed steps on the textured 1M-triangle scene, emitter gradients on"""
    import torch
    res, spp = 512, 64
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, textured=True)
    d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True}
    scene = mi.load_dict(d)
    integ = scene.integrator()
    grad_in = torch.full((res, res, 3), 1.0 / (res * res * 3), device="cuda")
    g = [mi.render_backward_distributed(scene, grad_in, integ, seed=1, spp=spp) for _ in range(4)]
    torch.cuda.synchronize()
    for k in g[0]:
        a, b = g[0][k].cpu().numpy(), g[-1][k].cpu().numpy()
        assert np.isfinite(b).all() and np.abs(b).max() > 0
        assert rel_l2(b, a) < 1e-4, k           # run-to-run: float atomics over ~10^8 terms in arbitrary order


# ------------------------------------------------------------------ N1 (i): forward parity on the 1M-triangle scenes at the bench's resolution

@pytest.mark.parametrize("flatten", [False, True], ids=["instanced1m", "flat1m"])
def test_forward_parity_1m_scene_512(mi, O, flatten):
    """512 x 512 film of the bench, 4 spp (1 M paths: seconds for the oracle on the box's host cores): image <= 1e-4, path / vertex counts equal"""
    res, spp = 512, 4
    scene, osc, sensor = _scene_1m(mi, O, res, spp, flatten=flatten)
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    gst = scene.integrator().stats()
    assert gst["paths"] == st.paths == res * res * spp and gst["vertices"] == st.vertices, (gst, st.paths, st.vertices)


def test_materials_1m_scene_512_forward_and_gradients(mi, O):
    """the flattened 1M-triangle scene with every BSDF model of the path (rough plastic walls; GGX conductor, diffuse and glass spheres) at the
    bench's film size: forward image <= 1e-4 with equal path / vertex counts (the material-sorted generic shading kernel at 1 M lanes), and
    the PRB gradients of the colours AND of alpha / eta / k / specular_reflectance <= 1e-3 at 256 x 256 x 8 spp"""
    res, spp = 512, 4
    scene, osc, sensor = _scene_1m(mi, O, res, spp, flatten=True, materials=True)      # the oracle's own lowering, BSDF records from the plugins' documented defaults
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    gst = scene.integrator().stats()
    # the rough models sample through erf / erfinv / exp / log / sincos: the product and the oracle each restate Dr.Jit's polynomial versions (har_math.h,
    # orc_math.h), so these are the SAME paths on the device as in the oracle, to the vertex, like the diffuse 1M scenes above
    assert gst["paths"] == st.paths and gst["vertices"] == st.vertices, (gst, st.vertices)
    res, spp = 256, 8
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=True, materials=True)
    d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5, "bsdf_parameter_gradients": True}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(6).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res * 3)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=11, spp=spp)
    gx, g_refl = osc.render_prb_backward_bsdf_params(sensor, grad_in, seed=11, spp=spp, max_depth=8)
    for key, (what, b) in scene._bsdf_param_keys().items():
        if what == "ior":
            continue
        rec = gx[b.index]
        want = {"alpha": rec[0:2].sum().reshape(1), "alpha_u": rec[0].sum(keepdims=True), "alpha_v": rec[1].sum(keepdims=True), "eta": rec[2], "k": rec[3], "slot1": rec[4]}[what]
        assert np.abs(want).max() > 0 and rel_l2(grads[key].cpu().numpy().reshape(-1), want.reshape(-1)) < 1e-3, key
    for key, (kind, b) in scene._param_keys().items():
        if kind == "rgb" and np.abs(g_refl[b.index]).max() > 0:
            assert rel_l2(grads[key].cpu().numpy().reshape(-1), g_refl[b.index]) < 1e-3, key


def test_material_queues_equal_the_generic_kernel(mi, O):
    """per-material shading queues (har_integrator_set_material_queues: k_classify + one k_shade launch per BSDF model) against the default generic
    kernel and the oracle: same paths, same vertex count, images equal up to the order of the film's float atomics; `path` and the primal pass of
    `prb` (whose hits feed the replay cache), on a scene with four models + a twosided conductor + escaping rays"""
    res, spp = 192, 8
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=True, materials=True, grid=6, n_u=24, n_v=12)
    d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.2, 0.3, 0.5]}}
    d.pop("ceiling", None)                                                   # let paths escape: the miss class
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    ref, st = osc.render_path(sensor, seed=3, spp=spp, max_depth=8)
    imgs = {}
    for on in (False, True):
        integ = mi.load_dict({"type": "path", "max_depth": 8, "material_queues": on})
        imgs[on] = mi.render(scene, integrator=integ, spp=spp, seed=3).cpu().numpy()
        gst = integ.stats()
        assert gst["paths"] == st.paths and gst["vertices"] == st.vertices
        assert rel_l2(imgs[on], ref) < 1e-4
    assert rel_l2(imgs[True], imgs[False]) < 1e-6
    grad_in = np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res * 3)
    g = {}
    for on in (False, True):
        integ = mi.load_dict({"type": "prb", "max_depth": 6, "material_queues": on})
        g[on] = integ.render_backward(scene, None, grad_in, seed=5, spp=spp)
    for k in g[False]:
        a, b = g[False][k].cpu().numpy(), g[True][k].cpu().numpy()
        if np.abs(a).max() > 0:
            assert rel_l2(b, a) < 1e-4, k


# ------------------------------------------------------------------ N1 (ii): PRB gradients on the instanced scene with a bitmap albedo

def test_prb_gradients_instanced_textured(mi, O):
    """instanced 1M-triangle scene whose `white` BSDF (walls + all spheres) carries the 256 x 256 bitmap albedo: texel gradients (atomics under
    contention: ~100 instances share the texels), constant-albedo and emitter-radiance gradients vs the oracle at 256 x 256 x 16 spp"""
    res, spp = 256, 16
    scene, osc, sensor = _scene_1m(mi, O, res, spp, textured=True, integrator={"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True})
    rng = np.random.default_rng(5)
    grad_in = rng.uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res * 3)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=7, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=7, spp=spp, max_depth=8)
    _check_prb_gradients(grads, g_refl, g_tex, g_emit)


def test_prb_gradients_flat1m_textured(mi, O):
    """the FLATTENED 1M-triangle scene (64 MB BVH) with the bitmap albedo on every third sphere and the walls: texel, constant-albedo and
    emitter-radiance gradients vs the oracle at 256 x 256 x 16 spp (round-2 verdict: flat1m had no PRB-gradient test)"""
    res, spp = 256, 16
    scene, osc, sensor = _scene_1m(mi, O, res, spp, flatten=True, textured=True, integrator={"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True})
    grad_in = np.random.default_rng(15).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res * 3)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=3, spp=spp, max_depth=8)
    _check_prb_gradients(grads, g_refl, g_tex, g_emit)


def test_forward_parity_c3_1024(mi, O):
    """BASELINE config 3 at its own film size: the instanced 1M-triangle scene, 1024 x 1024, 2 spp (2 M paths for the oracle): image <= 1e-4,
    path and vertex counts equal"""
    res, spp = 1024, 2
    scene, osc, sensor = _scene_1m(mi, O, res, spp)
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    gst = scene.integrator().stats()
    assert gst["paths"] == st.paths == res * res * spp and gst["vertices"] == st.vertices, (gst, st.paths, st.vertices)


# ------------------------------------------------------------------ the literal bench configuration (slow: ~2 minutes of oracle on the box's host cores)

@pytest.mark.slow
def test_bench_configuration_forward_parity_full_size(mi, O):
    """`bench.py`'s forward workload, literally: instanced 1M-triangle scene, 512 x 512 x 256 spp, max_depth 8, rr_depth 5, seed 0 -- one 2^26-lane
    wavefront, the size at which `k_splat` runs its one-pixel-per-block path and every bounce's queue is 10^7..10^8 entries.  Image <= 1e-4 and
    the per-frame integer counters (paths, path vertices) EQUAL to the oracle's; the oracle scene comes from the oracle's own lowering"""
    res, spp = 512, 256
    scene, osc, sensor = _scene_1m(mi, O, res, spp)
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    gst = scene.integrator().stats()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    assert gst["paths"] == st.paths == res * res * spp and gst["vertices"] == st.vertices, (gst, st.paths, st.vertices)


@pytest.mark.slow
def test_bench_configuration_prb_parity_full_film(mi, O):
    """`bench.py`'s PRB workload at the bench's film size: textured instanced 1M-triangle scene, 512 x 512, 64 spp (16.8 M paths; the bench's 256 spp
    would be 2 minutes of oracle), emitter gradients on: texel / constant-albedo / radiance gradients <= 1e-3"""
    res, spp = 512, 64
    scene, osc, sensor = _scene_1m(mi, O, res, spp, textured=True, integrator={"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True})
    grad_in = np.random.default_rng(21).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res * 3)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=9, spp=spp)
    g_refl, g_tex, g_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=9, spp=spp, max_depth=8)
    _check_prb_gradients(grads, g_refl, g_tex, g_emit)


# ------------------------------------------------------------------ N1 (iii): BASELINE config 4 at its own size

def test_config4_prb_texture_gradient_full_size(mi, O):
    """C4 (BASELINE.json configs[3]): Cornell box, 256 x 256 bitmap albedo, prb max_depth 6, 256 x 256 film x 256 spp, loss = mean(img^2):
    image and the 256 x 256 x 3 gradient tensor vs the oracle (16.7 M paths: tens of seconds on the box's host cores)"""
    import torch
    res, spp = 256, 256
    d = mi.textured_cornell_box(res=res, tex_res=256, spp=spp)
    scene = mi.load_dict(d)
    sd, sensor = O.cornell_box(res, res, white_texture=d["white"]["reflectance"]["data"])
    osc = O.OracleScene(sd)
    params = mi.traverse(scene)
    key = "white.reflectance.data"
    params[key].requires_grad_()
    img = mi.render(scene, params, spp=spp, seed=0)
    (img ** 2).mean().backward()
    g = params[key].grad.cpu().numpy()
    ref_img, _ = osc.render_prb(sensor, seed=0, spp=spp, max_depth=6)
    assert rel_l2(img.detach().cpu().numpy(), ref_img) < 1e-4
    grad_in = 2.0 * ref_img / ref_img.size
    _, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=mi.sample_tea_32(0, 1)[0], spp=spp, max_depth=6)
    assert g.shape == (256, 256, 3)
    assert rel_l2(g, g_tex[0]) < 1e-3


# ------------------------------------------------------------------ N1 (iv): BASELINE config 2 at its own film size, full parity

def test_config2_cornell_512_full_parity(mi, O):
    """C2 (BASELINE.json configs[1]) film 512 x 512, 16 spp: image <= 1e-4 and equal path / vertex counts (the 256 spp run of
    test_full_size_properties checks the size-independent properties)"""
    res, spp = 512, 16
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    scene = mi.load_dict(d)
    sd, sensor = O.cornell_box(res, res)
    osc = O.OracleScene(sd)
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
    assert rel_l2(img, ref) < 1e-4
    gst = scene.integrator().stats()
    assert gst["paths"] == st.paths and gst["vertices"] == st.vertices


@pytest.mark.parametrize("res,spp,rfilter", [(48, 256, "gaussian"), (40, 512, "gaussian"), (56, 64, "gaussian"), (33, 96, "gaussian"), (40, 256, "tent"), (40, 256, "box")])
def test_splat_gather_at_high_sample_counts(mi, O, res, spp, rfilter):
    """k_splat's LDS gather at the bench's samples per pixel: a 256-lane block holds exactly one pixel (256 spp), half a pixel (512), four
    pixels (64) or a ragged 2.67 (96, width x spp not a multiple of 256 -> some blocks span two rows and scatter) -- the tap loop of a tile column
    only visits the taps whose pixel has lanes in the block.  Film (RGB and the accumulated weights) vs the oracle's ImageBlock::put"""
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    d["sensor"]["film"]["rfilter"] = {"type": rfilter}
    d["integrator"]["max_depth"] = 3
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    film = scene.integrator().render_film(scene, seed=2, spp=spp).cpu().numpy()
    raw, _ = osc.render_path(sensor, seed=2, spp=spp, max_depth=3, raw=True)
    assert rel_l2(film[..., 3], raw[..., 3]) < 2e-6            # the weight channel: pure filter arithmetic
    assert rel_l2(film[..., :3], raw[..., :3]) < 1e-4


# ------------------------------------------------------------------ texel-gradient queues (TexelQueues, har_kernels.h)

@pytest.mark.parametrize("tex_res", [2, 7, 64, 512])
def test_texel_queue_gradients_all_texture_sizes(mi, O, tex_res):
    """the band-queue path of the PRB adjoint for textures from 2 x 2 (every path of the chip adds to the same four texels; bilinear wrap in both
    directions on every tap) over a non-power-of-two size to 512 x 512 (57 bands of 64 KB): texel gradients vs the oracle, and the queued path
    vs the direct atomics (HAR_TEXEL_QUEUES=0 is read once per process, so the comparison uses the replay cache switch, which disables the
    in-place commit and with it the queues)"""
    res, spp = 128, 16
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=16, n_v=8, textured=True, tex_res=tex_res)
    d["integrator"] = {"type": "prb", "max_depth": 6, "rr_depth": 5}
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    grad_in = np.random.default_rng(tex_res).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32) / (res * res)
    g = scene.integrator().render_backward(scene, None, grad_in, seed=3, spp=spp)["white.reflectance.data"].cpu().numpy()
    _, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=3, spp=spp, max_depth=6)
    assert rel_l2(g, g_tex[0]) < 1e-3
    d["integrator"]["replay_cache"] = False
    scene2 = mi.load_dict(d)
    g2 = scene2.integrator().render_backward(scene2, None, grad_in, seed=3, spp=spp)["white.reflectance.data"].cpu().numpy()
    assert rel_l2(g, g2) < 5e-4           # two float32 summation orders of up to ~10^6 terms per texel (2 x 2 texture)
