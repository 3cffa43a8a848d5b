#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run from the repo root IN THE BUILD CONTAINER
(`python tests/golden/make_golden.py`); the outputs are committed because neither
/root/reference nor a real mitsuba install exists on the GPU box.

Two kinds of fixture:

1. reference_kats.json -- known-answer vectors TRANSCRIBED from the reference's own test
   files under /root/reference (parsed, not retyped; each entry carries file:line).  These
   pin the oracle.  The reference itself cannot be imported here (no drjit),
   so these are the only reference-originated numbers available.

2. oracle_fixtures.npz -- outputs of the CPU oracle (oracle/libmi_oracle.so) on small seeded
   inputs: forward images, a PRB texture gradient, ray-query results, sampler streams.  The
   `-m gpu` tests compare the HIP path against them (in addition to running the oracle
   live), and a CPU test checks the oracle still reproduces them (drift guard).
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _lines(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read().splitlines()


def transcribe_kats():
    out = {"_comment": "transcribed by tests/golden/make_golden.py from the reference test-suite; do not edit by hand",
           "reference_version": "mitsuba3 v3.9.1 (include/mitsuba/mitsuba.h:11-13)"}
    # --- TEA (src/core/tests/test_random.py)
    rel = "src/core/tests/test_random.py"
    tea = {"float32": [], "float64": []}
    for no, line in enumerate(_lines(rel), 1):
        m = re.search(r"mi\.sample_tea_(float32|float64)\((\d+), (\d+), (\d+)\) == ([0-9.eE+-]+)", line)
        if m:
            tea[m.group(1)].append({"v0": int(m.group(2)), "v1": int(m.group(3)), "rounds": int(m.group(4)),
                                    "value": float(m.group(5)), "src": "%s:%d" % (rel, no)})
    assert len(tea["float32"]) == 8 and len(tea["float64"]) == 8
    out["tea"] = tea
    # --- Cornell pixel (src/integrators/tests/test_integrators.py: test02_path_directly_visible)
    rel = "src/integrators/tests/test_integrators.py"
    L = _lines(rel)
    crop = {}
    for no, line in enumerate(L, 1):
        m = re.search(r"\['film'\]\['(crop_offset_x|crop_offset_y|crop_width|crop_height)'\] = (\d+)", line)
        if m and no < 60:
            crop[m.group(1)] = int(m.group(2))
        m = re.search(r"dr\.allclose\(img\.array, \[([0-9., ]+)\]\)", line)
        if m and "cornell_pixel" not in out:
            out["cornell_pixel"] = {"crop": crop, "max_depth": 1, "value": [float(x) for x in m.group(1).split(",")],
                                    "rtol": 1e-5, "src": "%s:%d" % (rel, no)}
    assert out["cornell_pixel"]["crop"] == {"crop_offset_x": 124, "crop_offset_y": 36, "crop_width": 1, "crop_height": 1}
    # --- Cornell box constants (src/python/python/util.py: cornell_box())
    rel = "src/python/python/util.py"
    consts = {}
    for no, line in enumerate(_lines(rel), 1):
        m = re.search(r"'value': \[([0-9., ]+)\]", line)
        if m and 560 < no < 710:
            consts.setdefault("rgb_values", []).append({"value": [float(x) for x in m.group(1).split(",")], "src": "%s:%d" % (rel, no)})
        m = re.search(r"'fov': ([0-9.]+)", line)
        if m and 560 < no < 710:
            consts["fov"] = {"value": float(m.group(1)), "src": "%s:%d" % (rel, no)}
    out["cornell_constants"] = consts
    # --- stairs analytic depth (src/render/tests/test_kdtrees.py)
    out["stairs"] = {"n_steps": 20, "grid": 128, "formula": "t = 2 - floor(y*20)/20", "src": "src/render/tests/test_kdtrees.py:8-81"}
    # --- diffuse closed form (src/bsdfs/tests/test_diffuse.py:16-39)
    out["diffuse"] = {"reflectance": 0.5, "n": 20, "formula": "eval = 0.5/pi*cos(theta_o), pdf = cos(theta_o)/pi",
                      "src": "src/bsdfs/tests/test_diffuse.py:16-39"}
    # --- TEA constants + epsilon used by the path (include/mitsuba/core/random.h:76-90, math.h:17-22)
    rel = "include/mitsuba/core/random.h"
    keys = []
    for no, line in enumerate(_lines(rel), 1):
        if 70 < no < 95:
            keys += [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{8}", line)]
    out["tea_constants"] = {"values": keys, "src": "%s:76-90" % rel}
    # --- dielectric sample KATs (src/bsdfs/tests/test_dielectric.py: example_bsdf + test02_sample / test03_sample_reverse, Radiance mode)
    rel = "src/bsdfs/tests/test_dielectric.py"
    L = _lines(rel); txt = "\n".join(L)
    m = re.search(r"def example_bsdf\(reflectance=([0-9.]+), transmittance=([0-9.]+)\)", txt)
    refl, trans = float(m.group(1)), float(m.group(2))
    int_ior = float(re.search(r"'int_ior': ([0-9.]+)", txt).group(1)); ext_ior = float(re.search(r"'ext_ior': ([0-9.]+)", txt).group(1))
    pdf_r = float(re.search(r"dr\.allclose\(bs\.pdf, ([0-9.]+)\)", txt).group(1))
    line = next(i for i, l in enumerate(L, 1) if "def test02_sample" in l)
    out["dielectric_sample"] = {
        "bsdf": {"specular_reflectance": refl, "specular_transmittance": trans, "int_ior": int_ior, "ext_ior": ext_ior},
        "cases": [
            {"wi": [0, 0, 1], "sample1": 0.0, "weight": [refl] * 3, "pdf": pdf_r, "eta": 1.0, "wo": [0, 0, 1], "delta": True},
            {"wi": [0, 0, 1], "sample1": 0.05, "weight": [trans / int_ior ** 2] * 3, "pdf": 1 - pdf_r, "eta": int_ior, "wo": [0, 0, -1], "delta": True},
            {"wi": [0, 0, -1], "sample1": 0.0, "weight": [refl] * 3, "pdf": pdf_r, "eta": 1.0, "wo": [0, 0, -1], "delta": True},
            {"wi": [0, 0, -1], "sample1": 0.05, "weight": [trans * int_ior ** 2] * 3, "pdf": 1 - pdf_r, "eta": 1 / int_ior, "wo": [0, 0, 1], "delta": True},
        ],
        "src": "%s:%d-%d (TransportMode.Radiance branch)" % (rel, line, line + 62)}
    # --- dielectric under a BSDFContext (same file: test02/test03 Importance branch, test04_sample_specific_component): mode 0 Radiance / 1 Importance,
    #     type_mask / component select a lobe -- then pdf = 1 and the weight carries the Fresnel term (0.3 * 0.04, 0.6 * (1 - 0.04) [/ 1.5^2 under Radiance])
    l4 = next(i for i, l in enumerate(L, 1) if "def test04_sample_specific_component" in l)
    assert "0.3 * 0.04" in txt and "0.6 * (1 - 0.04) / 1.5**2" in txt and "ctx.component = 3" in txt
    DR, DT, ALL, NONE = 0x20, 0x40, 0x1ff, 0xffffffff
    ctx_cases = []
    for mode in (1, 0):                                          # i == 0: Importance, i == 1: Radiance
        t_w = trans if mode == 1 else trans / int_ior ** 2
        ctx_cases += [   # test02_sample (both lobes): wi = +z
            {"ctx": [mode, ALL, NONE], "wi": [0, 0, 1], "sample1": 0.0, "weight": [refl] * 3, "pdf": pdf_r, "eta": 1.0, "wo": [0, 0, 1], "type": DR, "component": 0},
            {"ctx": [mode, ALL, NONE], "wi": [0, 0, 1], "sample1": 0.05, "weight": [t_w] * 3, "pdf": 1 - pdf_r, "eta": int_ior, "wo": [0, 0, -1], "type": DT, "component": 1},
            # test03_sample_reverse: wi = -z
            {"ctx": [mode, ALL, NONE], "wi": [0, 0, -1], "sample1": 0.0, "weight": [refl] * 3, "pdf": pdf_r, "eta": 1.0, "wo": [0, 0, -1], "type": DR, "component": 0},
            {"ctx": [mode, ALL, NONE], "wi": [0, 0, -1], "sample1": 0.05, "weight": [trans if mode == 1 else trans * int_ior ** 2] * 3, "pdf": 1 - pdf_r,
             "eta": 1 / int_ior, "wo": [0, 0, 1], "type": DT, "component": 1}]
        for sample in (0.0, 0.5, 1.0):                           # test04_sample_specific_component
            for sel in (0, 1):
                ctx_cases.append({"ctx": [mode, DR, NONE] if sel == 0 else [mode, ALL, 0], "wi": [0, 0, 1], "sample1": sample, "weight": [refl * pdf_r] * 3, "pdf": 1.0,
                                  "eta": 1.0, "wo": [0, 0, 1], "type": DR, "component": 0})
                ctx_cases.append({"ctx": [mode, DT, NONE] if sel == 0 else [mode, ALL, 1], "wi": [0, 0, 1], "sample1": sample,
                                  "weight": [trans * (1 - pdf_r) * (1.0 if mode == 1 else 1 / int_ior ** 2)] * 3, "pdf": 1.0, "eta": int_ior, "wo": [0, 0, -1], "type": DT, "component": 1})
    ctx_cases.append({"ctx": [0, ALL, 3], "wi": [0, 0, 1], "sample1": 0.0, "weight": [0.0] * 3, "zero_only": True})      # ctx.component = 3: nothing enabled
    out["dielectric_context"] = {"bsdf": out["dielectric_sample"]["bsdf"], "cases": ctx_cases, "src": "%s:%d-%d and :%d-%d" % (rel, line, line + 62, l4, l4 + 44)}
    # --- twosided(diffuse) pdf (src/bsdfs/tests/test_twosided.py: test02_pdf)
    rel = "src/bsdfs/tests/test_twosided.py"
    line = next(i for i, l in enumerate(_lines(rel), 1) if "def test02_pdf" in l)
    out["twosided_pdf"] = {"wi": [0, 0, 1], "cases": [{"wo": [0, 0, 1], "pdf": 1 / np.pi}, {"wo": [0, 0, -1], "pdf": 0.0}], "src": "%s:%d-%d" % (rel, line, line + 18)}
    # --- published PCG32 demo vector (pcg-random.org pcg32-demo, seed 42/54); drjit is NOT in the tree
    out["pcg32_published"] = {"initstate": 42, "initseq": 54,
                              "outputs": [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e],
                              "src": "pcg-random.org pcg32-demo (drjit 1.5.0 dr::PCG32 is not in /root/reference: parity unpinned)"}
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return out


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.9, 0.9, (3, n)).astype(np.float32)
    d = rng.normal(size=(3, n)).astype(np.float32)
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    return o, d.astype(np.float32)


def checker_texture(res):
    """0.5 + 0.25 * checker(8 x 8), H x W x 3 f32"""
    i = (np.arange(res) * 8 // res)
    c = ((i[:, None] + i[None, :]) & 1).astype(np.float32)
    return np.repeat((0.5 + 0.25 * (2 * c - 1))[:, :, None], 3, axis=2).astype(np.float32)


def oracle_fixtures():
    from oracle import oracle as O
    fx = {}
    # forward path, Cornell 32x32, 8 spp, seed 0, max_depth 8 (gaussian) and box filter + crop
    sd, sensor = O.cornell_box(32, 32)
    osc = O.OracleScene(sd)
    fx["cornell32_spp8_seed0_path"], _ = osc.render_path(sensor, seed=0, spp=8, max_depth=8)
    fx["cornell32_spp8_seed0_prb"], _ = osc.render_prb(sensor, seed=0, spp=8, max_depth=6)
    # ray queries (closest hit + shadow) on seeded rays
    n = 4096
    o, d = random_rays(n, 1)
    maxt = np.full(n, 3.402823466e+38, np.float32)
    t, u, v, prim, shape, inst = osc.ray_intersect(o, d, maxt, naive=True)
    fx["rays_o"], fx["rays_d"] = o, d
    fx["cornell_hit_t"], fx["cornell_hit_u"], fx["cornell_hit_v"] = t, u, v
    fx["cornell_hit_prim"], fx["cornell_hit_shape"] = prim, shape
    fx["cornell_ray_test_maxt1"] = osc.ray_test(o, d, np.full(n, 1.0, np.float32))
    # sampler streams: seed 7, lanes 0..15, 5 draws
    st = np.empty((16, 5), np.float32)
    for lane in range(16):
        out = np.empty(5, np.float32); O.lib().orc_sampler_stream(7, lane, 5, O.fp(out)); st[lane] = out
    fx["sampler_seed7"] = st
    # PRB texture gradient (C4 at fixture size): 24x24 film, 8x8 texture, 8 spp, loss = mean(img^2)
    tex = checker_texture(8)
    sd, sensor = O.cornell_box(24, 24, white_texture=tex)
    osc = O.OracleScene(sd)
    img, _ = osc.render_prb(sensor, seed=0, spp=8, max_depth=6)
    grad_in = (2.0 * img / img.size).astype(np.float32)
    g_refl, g_tex, _ = osc.render_prb_backward(sensor, grad_in, seed=0x1234, spp=8, max_depth=6)
    fx["c4_texture"] = tex; fx["c4_img"] = img; fx["c4_grad_in"] = grad_in
    fx["c4_grad_refl"] = g_refl; fx["c4_grad_tex"] = g_tex[0]
    # vertex-position gradients of prb (slab scene of tests/test_shape_gradients_cpu.py: floor + ceiling, 16 x 16, 16 spp, seed 3, max_depth 5)
    import mitsuba3_amd as mi
    from tests.test_cpu_host import oracle_scene_from
    from tests.test_shape_gradients_cpu import slab_scene, mesh_index
    mi.set_variant("hip_ad_rgb")
    scene = mi.load_dict(slab_scene(mi, 16, textured=True))
    osc, sensor = oracle_scene_from(O, scene)
    ids = [mesh_index(scene, n) for n in ("floor", "ceiling")]
    w = np.random.default_rng(4).uniform(0.5, 1.5, (16, 16, 3)).astype(np.float32)
    g_pos, _, _, _ = osc.render_prb_backward_shape(sensor, w, ids, seed=3, spp=16, max_depth=5)
    fx["shape_grad_in"] = w; fx["shape_grad_floor"] = g_pos[ids[0]]; fx["shape_grad_ceiling"] = g_pos[ids[1]]
    # round-4 plugins in ONE scene: the Cornell box lit by its area light, a point light, a spot light and a directional light, seen through an orthographic camera
    # (tests/test_golden_cpu.py: round4_scene); 32 x 32, 8 spp, seed 2: forward image (path), prb image and the gradients w.r.t. the four emitters' parameters
    from tests.test_golden_cpu import round4_scene
    scene = mi.load_dict(round4_scene(mi))
    osc, sensor = O.scene_from_product(scene)
    fx["r4_path"], _ = osc.render_path(sensor, seed=2, spp=8, max_depth=8)
    fx["r4_prb"], _ = osc.render_prb(sensor, seed=2, spp=8, max_depth=6)
    w4 = np.random.default_rng(11).uniform(0.5, 1.5, (32, 32, 3)).astype(np.float32)
    g_refl4, _, g_emit4, _ = osc.render_prb_backward_emitters(sensor, w4, seed=5, spp=8, max_depth=6)
    fx["r4_grad_in"] = w4; fx["r4_grad_refl"] = g_refl4; fx["r4_grad_emit"] = g_emit4
    # round 5: the Cornell box lit by a BITMAP on its ceiling rectangle (emitter type 7; tests/test_textured_area_light_cpu.py: lit_box / _bitmap), 24 x 24, 16 spp, seed 3:
    # forward image, prb image, and 4096 (uv, pdf) pairs of BitmapTexture::sample_position on a fixed grid of samples
    from tests.test_textured_area_light_cpu import lit_box, _bitmap
    scene = mi.load_dict(lit_box(mi, _bitmap(5), 24, wrap_mode="mirror"))
    osc, sensor = O.scene_from_product(scene)
    fx["texlight_path"], _ = osc.render_path(sensor, seed=3, spp=16, max_depth=6)
    fx["texlight_prb"], _ = osc.render_prb(sensor, seed=3, spp=16, max_depth=5)
    g = (np.arange(64, dtype=np.float32) + 0.5) / 64
    s = np.ascontiguousarray(np.stack(np.meshgrid(g, g), -1).reshape(-1, 2), np.float32)
    uv = np.zeros((len(s), 2), np.float32); pdf = np.zeros(len(s), np.float32)
    import ctypes as C
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert O.lib().orc_emitter_texture_sample_position(osc.handle, 0, vp(s), len(s), vp(uv), vp(pdf), 0) == 0
    fx["texlight_sample_uv"] = uv; fx["texlight_sample_pdf"] = pdf
    # the film is accumulated by several threads in an order that differs from run to run (1e-7 relative): arrays that are already committed stay as committed unless
    # they moved by more than that, so that regenerating the file only ADDS what is new
    path = os.path.join(HERE, "oracle_fixtures.npz")
    if os.path.exists(path):
        for k, v in dict(np.load(path)).items():
            if k in fx and v.shape == fx[k].shape and (np.array_equal(v, fx[k]) or (v.dtype.kind == "f" and np.abs(v - fx[k]).max() <= 2e-6 * max(float(np.abs(v).max()), 1e-30))):
                fx[k] = v
    np.savez_compressed(path, **fx)
    return fx


if __name__ == "__main__":
    old = dict(np.load(os.path.join(HERE, "oracle_fixtures.npz"))) if os.path.exists(os.path.join(HERE, "oracle_fixtures.npz")) else {}
    k = transcribe_kats()
    print("reference_kats.json:", sorted(k.keys()))
    f = oracle_fixtures()
    print("oracle_fixtures.npz:", {n: v.shape for n, v in f.items()})
    changed = [n for n in old if n in f and not np.array_equal(old[n], f[n])]
    print("fixtures that CHANGED w.r.t. the previous file (must be empty unless the oracle was changed on purpose):", changed)
