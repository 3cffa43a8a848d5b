import sys, time, ctypes as C; sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
d = mi.instanced_spheres_scene(width=64, height=64, spp=4, grid=10, n_u=100, n_v=50, flatten=True); scene = mi.load_dict(d)
H = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests', 'host_harness', 'libhost_harness.so')); H.hh_scene_create.restype = C.c_void_p
err = C.create_string_buffer(256); desc = scene.desc(); t2=time.time(); h = H.hh_scene_create(C.byref(desc), err, 256); t3=time.time()
print("lower_scene %.2fs" % (t3-t2), scene.meshes[0]["V"].shape if scene.meshes else None)
