"""where does the forward image of the 1M-triangle scene differ from the oracle?  (diagnostic; run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitsuba3_amd as mi
from oracle import oracle as O
mi.set_variant("hip_ad_rgb")
flatten = len(sys.argv) > 1 and sys.argv[1] == "flat"
res, spp = 512, 4
scene = mi.load_dict(mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=flatten))
osc, sensor = O.scene_from_product(scene)
img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
gst = scene.integrator().stats()
ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8)
print("stats gpu", gst, "oracle paths", st.paths, "vertices", st.vertices)
d = (img.astype(np.float64) - ref.astype(np.float64))
print("rel_l2", np.linalg.norm(d) / np.linalg.norm(ref))
e = np.abs(d).sum(-1)
idx = np.argsort(e.ravel())[::-1][:12]
for k in idx:
    y, x = divmod(int(k), res)
    print("px", x, y, "err", e[y, x], "gpu", img[y, x], "ref", ref[y, x])
tot = (d ** 2).sum()
srt = np.sort((d ** 2).sum(-1).ravel())[::-1]
print("share of squared error in top 10 / 100 / 1000 pixels:", srt[:10].sum() / tot, srt[:100].sum() / tot, srt[:1000].sum() / tot, "pixels with err > 1e-6:", int((e > 1e-6).sum()))
# ray queries: random rays + camera-like rays, bit-exact?
rng = np.random.default_rng(3); n = 400000
o = rng.uniform(-0.9, 0.9, (3, n)).astype(np.float32); dd = rng.normal(size=(3, n)).astype(np.float32); dd /= np.linalg.norm(dd, axis=0); dd = dd.astype(np.float32)
maxt = np.full(n, 3.402823466e+38, np.float32)
rr = osc.ray_intersect(o, dd, maxt)
pi = scene.ray_intersect_preliminary(mi.Ray3f(o, dd, maxt))
t = pi.t.cpu().numpy()
print("ray t mismatches:", int((t != rr[0]).sum()), "of", n, " prim mismatches:", int((pi.prim_index.cpu().numpy().astype(np.uint32)[np.isfinite(rr[0])] != rr[3][np.isfinite(rr[0])]).sum()))
