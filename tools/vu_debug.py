import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
scene = mi.load_dict(mi.instanced_spheres_scene(width=64, height=64, spp=4))
mi.render(scene, spp=4, seed=0)
params = mi.traverse(scene)
key = "spheres.ball.positions"
m = scene._position_keys()[key]
t = params[key]
L = mi.lib()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = L.har_scene_update_vertices_device(scene._h, int(m), t.data_ptr(), None)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("instanced mesh (%d vertices, 100 instances): host %.3f ms, + gpu drain %.3f ms rc %d" % (t.numel() // 3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc))
os.environ["HAR_HOST_TLAS_UPDATE"] = "1"
