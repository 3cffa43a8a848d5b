"""Are paths through rough BSDFs (and environment-map lookups) the SAME paths on the device and in the oracle?  Renders a few scenes with `path` on both
and prints path / vertex counts and the image difference.  With the restated elementary functions (har_math.h exp_ / log_ / erf_ / atan2_ / acos_ /
tan_ == orc_math.h) the vertex counts must be equal; with libm on one side and the device library on the other about 1e-5 of the paths differed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mitsuba3_amd as mi
from oracle import oracle as O

mi.set_variant("hip_ad_rgb")


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def run(name, d, spp, max_depth=8):
    scene = mi.load_dict(d)
    osc, sensor = O.scene_from_product(scene)
    img = mi.render(scene, spp=spp, seed=0).cpu().numpy()
    ref, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=max_depth)
    g = scene.integrator().stats()
    print(f"{name:34s} paths {g['paths']:9d} / {st.paths:9d}   vertices {g['vertices']:10d} / {st.vertices:10d}  diff {g['vertices'] - st.vertices:+d}   image rel_l2 {rel_l2(img, ref):.2e}", flush=True)


def box(bsdf, res=128, spp=16):
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = d["sensor"]["film"]["height"] = res
    d["sensor"]["sampler"] = {"type": "independent", "sample_count": spp}
    d["integrator"] = {"type": "path", "max_depth": 8, "rr_depth": 5}
    d["white"] = dict(bsdf)                 # walls and boxes refer to it
    return d


if __name__ == "__main__":
    models = {
        "roughplastic beckmann": {"type": "roughplastic", "distribution": "beckmann", "alpha": 0.2},
        "roughplastic ggx": {"type": "roughplastic", "distribution": "ggx", "alpha": 0.2},
        "roughconductor beckmann": {"type": "roughconductor", "distribution": "beckmann", "alpha": 0.15, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
        "roughconductor beckmann aniso": {"type": "roughconductor", "distribution": "beckmann", "alpha_u": 0.05, "alpha_v": 0.3, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
        "roughconductor ggx aniso": {"type": "roughconductor", "distribution": "ggx", "alpha_u": 0.05, "alpha_v": 0.3, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
        "roughconductor beckmann no-vis": {"type": "roughconductor", "distribution": "beckmann", "alpha": 0.15, "sample_visible": False, "eta": {"type": "rgb", "value": [0.2, 0.9, 1.1]}, "k": {"type": "rgb", "value": [3.9, 2.4, 2.2]}},
        "plastic": {"type": "plastic"},
    }
    for name, b in models.items():
        try:
            run(name, box(b), 16)
        except Exception as e:
            print(name, "FAILED", repr(e), flush=True)
    res, spp = 512, 4
    run("materials1m 512x512x4", mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=True, materials=True), spp)
