"""debug helper for tests/test_gpu_fuzz_parity.py::test_random_scene_options: one seed, the prb backward call under every option combination, against the oracle"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
from oracle import oracle as O
from tests.test_gpu_fuzz_parity import random_scene

seed = int(os.environ.get("SEED", "2183"))
d, cfg = random_scene(mi, seed + 500)
spp, md, rr = cfg["spp"], cfg["max_depth"], cfg["rr_depth"]
if spp == 12: spp = 8
def show(x, ind=0):
    for k, v in x.items():
        if isinstance(v, dict):
            print(" " * ind + k + ": " + str({a: (b if not isinstance(b, (dict, np.ndarray)) else "...") for a, b in v.items()}))
            if v.get("type") in ("twosided", "shapegroup") or "bsdf" in v: show({a: b for a, b in v.items() if isinstance(b, dict)}, ind + 4)
show(d)
print("cfg", cfg, "spp", spp)
grad_in = None
for chunk, mq, rc, bp in itertools.product((None, 2048), (False, True), (None, False), (False, True)):
    opts = {}
    if chunk: opts["chunk_lanes"] = chunk
    if mq: opts["material_queues"] = True
    if rc is False and not bp: opts["replay_cache"] = False
    d["integrator"] = dict({"type": "prb", "max_depth": md, "rr_depth": rr, "bsdf_parameter_gradients": bp}, **opts)
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    if grad_in is None:
        ref, _ = osc.render_prb(sensor, seed=seed, spp=spp, max_depth=md, rr_depth=rr)
        grad_in = np.random.default_rng(seed).uniform(0.5, 1.5, ref.shape).astype(np.float32)
        w_refl, w_tex, w_emit, _ = osc.render_prb_backward_emitters(sensor, grad_in, seed=seed + 2, spp=spp, max_depth=md, rr_depth=rr)
    grads = scene.integrator().render_backward(scene, None, grad_in, seed=seed + 2, spp=spp)
    line = []
    for k, (kind, b) in scene._param_keys().items():
        want = w_emit[b] if kind == "emit" else (w_tex[b.tex_index] if kind == "tex" else w_refl[b.index])
        got = grads[k].cpu().numpy()
        err = np.abs(got - want).max(); sc = np.abs(want).max()
        if err > 1e-3 * sc + 5e-7:
            line.append("%s got %s want %s" % (k, got.reshape(-1)[:3], np.asarray(want).reshape(-1)[:3]))
    print("chunk", chunk, "mq", mq, "replay_cache", rc, "bsdf_params", bp, "->", line or "ok")
