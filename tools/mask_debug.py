"""diagnostic: the FIRST har_integrator_sample of a process, with the allocator pool pre-filled with a chosen bit pattern (argv[1]: none | zero | one | nan | big | neg)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitsuba3_amd as mi
from oracle import oracle as O
mi.set_variant("hip_ad_rgb")
fill = sys.argv[1] if len(sys.argv) > 1 else "none"
use_mask = (sys.argv[2] if len(sys.argv) > 2 else "mask") == "mask"
n = 40000
rng = np.random.default_rng(13)
o = np.tile(np.array([[0.0], [0.0], [3.9]], np.float32), (1, n)); d = rng.normal(size=(3, n)).astype(np.float32); d[2] = -np.abs(d[2]) - 1.0
d /= np.linalg.norm(d, axis=0); d = np.ascontiguousarray(d, np.float32); maxt = np.full(n, 3.402823466e+38, np.float32)
if fill != "none":
    val = {"zero": 0.0, "one": 1.0, "nan": float("nan"), "big": 1e30, "neg": -1.0}[fill]
    junk = [torch.full((1 << 24,), val, device="cuda") for _ in range(8)]; torch.cuda.synchronize(); del junk
for it in range(2):
    dd = mi.cornell_box(); dd["integrator"] = {"type": "path", "max_depth": 5, "rr_depth": 3}
    scene = mi.load_dict(dd); osc, _ = O.scene_from_product(scene)
    active = np.random.default_rng(4).random(n) < 0.7
    sampler = mi.Sampler({"sample_count": 4, "seed": 2}); sampler.seed(1, n)
    spec, valid = scene.integrator().sample(scene, sampler, mi.Ray3f(o, d, maxt), active=active if use_mask else True)
    ref, rvalid, rstate = osc.integrator_sample(o, d, maxt, seed=3, max_depth=5, rr_depth=3, prb=False, active=active if use_mask else None)
    v = valid.cpu().numpy().astype(np.uint8); sp = spec.cpu().numpy()
    bad = np.nonzero(v != rvalid)[0]
    print(fill, "mask" if use_mask else "nomask", "call", it, "mismatches", bad.size, "rel l2", float(np.linalg.norm(sp - ref) / np.linalg.norm(ref)), "stats", scene.integrator().stats())
