"""ctypes binding of the CPU ORACLE (oracle/libmi_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (mitsuba3_amd/) never imports it.
Parity status: see oracle/mi_oracle.h ("parity unpinned" for Dr.Jit/Embree
arithmetic that is not in the reference tree).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)


class Mesh(C.Structure):
    _fields_ = [("vertex_ptr", c_f32p), ("index_ptr", c_u32p), ("vertex_count", C.c_uint32),
                ("face_count", C.c_uint32), ("bsdf", C.c_uint32), ("emitter", C.c_int32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class ShapeGroup(C.Structure):
    _fields_ = [("first_mesh", C.c_uint32), ("mesh_count", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("group", C.c_uint32), ("to_world", C.c_float * 12), ("to_object", C.c_float * 12)]


class BSDF(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_int32), ("reflectance", C.c_float * 3), ("flags", C.c_uint32),
                ("reflectance2", C.c_float * 3), ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("eta", C.c_float),
                ("eta_c", C.c_float * 3), ("k_c", C.c_float * 3), ("back", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("data", c_f32p), ("width", C.c_uint32), ("height", C.c_uint32), ("mode", C.c_uint32), ("reserved", C.c_uint32), ("to_uv", C.c_float * 6)]


class Emitter(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mesh", C.c_uint32), ("radiance", C.c_float * 3),
                ("to_world", C.c_float * 12), ("normal", C.c_float * 3), ("inv_area", C.c_float), ("to_local", C.c_float * 12), ("sampling_weight", C.c_float), ("radiance_texture", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(Mesh)), ("mesh_count", C.c_uint32), ("top_mesh_count", C.c_uint32),
                ("groups", C.POINTER(ShapeGroup)), ("group_count", C.c_uint32), ("pad0", C.c_uint32),
                ("instances", C.POINTER(Instance)), ("instance_count", C.c_uint32), ("pad1", C.c_uint32),
                ("bsdfs", C.POINTER(BSDF)), ("bsdf_count", C.c_uint32), ("pad2", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32), ("pad3", C.c_uint32),
                ("emitters", C.POINTER(Emitter)), ("emitter_count", C.c_uint32), ("pad4", C.c_uint32)]


class Sensor(C.Structure):
    _fields_ = [("sample_to_camera", C.c_float * 16), ("to_world", C.c_float * 16),
                ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_width", C.c_uint32), ("film_height", C.c_uint32),
                ("crop_offset_x", C.c_uint32), ("crop_offset_y", C.c_uint32),
                ("crop_width", C.c_uint32), ("crop_height", C.c_uint32),
                ("rfilter", C.c_uint32), ("rfilter_stddev", C.c_float), ("rfilter_param1", C.c_float),
                ("sample_border", C.c_uint32), ("principal_point_offset_x", C.c_float), ("principal_point_offset_y", C.c_float),
                ("projection", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("vertices", C.c_uint64), ("closest_rays", C.c_uint64),
                ("shadow_rays", C.c_uint64)]


def build():
    """Compile the oracle (g++); building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmi_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_reflectance.argtypes = [C.c_void_p, C.c_uint32, c_f32p]
        L.orc_scene_set_texture.argtypes = [C.c_void_p, C.c_uint32, c_f32p]
        L.orc_ray_intersect.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, C.c_int,
                                        c_f32p, c_f32p, c_f32p, c_u32p, c_u32p, c_u32p]
        L.orc_ray_test.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, C.c_int, C.POINTER(C.c_uint8)]
        for fn in (L.orc_render_path, L.orc_render_prb):
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.POINTER(Sensor), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                           C.c_uint64, C.c_uint64, c_f32p, C.POINTER(Stats), C.c_int]
        L.orc_render_path_passes.restype = C.c_int
        L.orc_render_path_passes.argtypes = [C.c_void_p, C.POINTER(Sensor), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                             C.c_uint64, C.c_uint64, c_f32p, C.POINTER(Stats), C.c_int]
        L.orc_render_path_scalar.restype = C.c_int
        L.orc_render_path_scalar.argtypes = [C.c_void_p, C.POINTER(Sensor), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32,
                                             c_f32p, C.POINTER(Stats), c_u32p]
        L.orc_hier2d_create.restype = C.c_void_p
        L.orc_hier2d_create.argtypes = [c_f32p, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_hier2d_destroy.argtypes = [C.c_void_p]
        for fn in (L.orc_hier2d_sample, L.orc_hier2d_invert):
            fn.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p]
        L.orc_hier2d_eval.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p]
        L.orc_hier2d_data.restype = C.c_uint32
        L.orc_hier2d_data.argtypes = [C.c_void_p, c_f32p, c_u32p, c_u32p]
        L.orc_envmap_create.restype = C.c_void_p
        L.orc_envmap_create.argtypes = [c_f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, c_f32p, c_f32p]
        L.orc_envmap_destroy.argtypes = [C.c_void_p]
        L.orc_envmap_set_bsphere.argtypes = [C.c_void_p, c_f32p, C.c_float]
        L.orc_envmap_eval.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p]
        L.orc_envmap_sample_direction.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_envmap_pdf_direction.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p]
        L.orc_render_prb_backward_ex.restype = C.c_int
        L.orc_render_prb_backward_ex.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, c_f32p,
                                                 C.POINTER(c_f32p), c_f32p, C.POINTER(Stats), C.c_int]
        L.orc_scene_set_emitter_radiance.argtypes = [C.c_void_p, C.c_uint32, c_f32p]
        L.orc_scene_set_hide_emitters.argtypes = [C.c_void_p, C.c_int]; L.orc_scene_set_hide_emitters.restype = None
        L.orc_render_prb_backward.restype = C.c_int
        L.orc_render_prb_backward.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, C.c_uint32, C.c_uint32,
                                              C.c_int32, C.c_int32, c_f32p, C.POINTER(c_f32p),
                                              C.POINTER(Stats), C.c_int]
        L.orc_film_develop.argtypes = [c_f32p, C.c_uint32, C.c_uint32, c_f32p]
        L.orc_sample_tea_32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, c_u32p]
        L.orc_sample_tea_float32.restype = C.c_float
        L.orc_sample_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.orc_sample_tea_float64.restype = C.c_double
        L.orc_sample_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        u64p = C.POINTER(C.c_uint64)
        L.orc_pcg32_seed.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.orc_pcg32_next_uint32.restype = C.c_uint32
        L.orc_pcg32_next_uint32.argtypes = [u64p]
        L.orc_pcg32_next_float32.restype = C.c_float
        L.orc_pcg32_next_float32.argtypes = [u64p]
        L.orc_sampler_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, c_f32p]
        L.orc_rfilter_eval.restype = C.c_float
        L.orc_rfilter_eval.argtypes = [C.c_uint32, C.c_float, C.c_float]
        L.orc_rfilter_eval2.restype = C.c_float
        L.orc_rfilter_eval2.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float]
        L.orc_film_put.argtypes = [C.POINTER(Sensor), C.c_uint32, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_sensor_sample_ray.argtypes = [C.POINTER(Sensor), C.c_uint32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_diffuse_eval_pdf.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_diffuse_sample.argtypes = [c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_square_to_cosine_hemisphere.argtypes = [c_f32p, c_f32p]
        L.orc_coordinate_system.argtypes = [c_f32p, c_f32p, c_f32p]
        L.orc_sincos.restype = C.c_float
        L.orc_sincos.argtypes = [C.c_float, c_f32p]
        L.orc_default_threads.restype = C.c_int
        L.orc_default_threads.argtypes = []
        L.orc_math_fn.restype = C.c_float
        L.orc_math_fn.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_surface_interaction.argtypes = [C.c_void_p, c_f32p, c_f32p, C.c_float, C.c_float, C.c_float,
                                              C.c_uint32, C.c_uint32, C.c_uint32, c_f32p]
        L.orc_look_at.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_translate.argtypes = [c_f32p, c_f32p]
        L.orc_scale.argtypes = [c_f32p, c_f32p]
        L.orc_rotate.argtypes = [c_f32p, C.c_float, c_f32p]
        L.orc_matmul.argtypes = [c_f32p, c_f32p, c_f32p]
        L.orc_affine_inverse.argtypes = [c_f32p, c_f32p]
        L.orc_perspective_sensor.argtypes = [c_f32p, C.c_double, C.c_char_p, C.c_float, C.c_float,
                                             C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint32, C.c_float, C.POINTER(Sensor)]
        L.orc_rectangle.argtypes = [c_f32p, c_f32p, c_u32p, c_f32p, c_f32p]
        L.orc_cube.argtypes = [c_f32p, c_f32p, c_u32p]
        L.orc_bake_mesh.argtypes = [c_f32p, c_f32p, C.c_uint32, c_u32p, C.c_uint32]; L.orc_bake_mesh.restype = None
        _LIB = L
    return _LIB


def fp(a):
    return a.ctypes.data_as(c_f32p)


def up(a):
    return a.ctypes.data_as(c_u32p)


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


# --------------------------------------------------------------------------
#  Transform helpers (32-float matrix + inverse-transpose pairs)
# --------------------------------------------------------------------------

class T:
    """Mirror of mi.ScalarTransform4f chaining, evaluated by the oracle."""

    def __init__(self, data=None):
        if data is None:
            data = np.concatenate([np.eye(4, dtype=np.float32).ravel()] * 2)
        self.data = f32(data)

    def _mul(self, other):
        out = np.empty(32, np.float32)
        lib().orc_matmul(fp(self.data), fp(other), fp(out))
        return T(out)

    def translate(self, v):
        out = np.empty(32, np.float32); lib().orc_translate(fp(f32(v)), fp(out)); return self._mul(out)

    def scale(self, v):
        v = f32([v, v, v]) if np.isscalar(v) else f32(v)
        out = np.empty(32, np.float32); lib().orc_scale(fp(v), fp(out)); return self._mul(out)

    def rotate(self, axis, angle):
        out = np.empty(32, np.float32); lib().orc_rotate(fp(f32(axis)), float(angle), fp(out)); return self._mul(out)

    def look_at(self, origin, target, up):
        out = np.empty(32, np.float32)
        lib().orc_look_at(fp(f32(origin)), fp(f32(target)), fp(f32(up)), fp(out)); return self._mul(out)

    def inverse(self):
        out = np.empty(32, np.float32); lib().orc_affine_inverse(fp(self.data), fp(out)); return T(out)

    @property
    def matrix(self):
        return self.data[:16].reshape(4, 4)

    def col_major_3x4(self):
        return np.ascontiguousarray(self.matrix[:3, :].T).ravel().astype(np.float32)


# --------------------------------------------------------------------------
#  Scene container: owns numpy arrays + the ctypes description
# --------------------------------------------------------------------------

class Hier2D:
    """Hierarchical2D<Float, 0> (distr_2d.h:370-860)"""
    def __init__(self, data, normalize=True):
        data = f32(data); assert data.ndim == 2
        self.h = lib().orc_hier2d_create(fp(data), data.shape[1], data.shape[0], int(normalize))
        assert self.h
    def __del__(self):
        if getattr(self, "h", None): lib().orc_hier2d_destroy(self.h)
    def _run(self, fn, x):
        x = f32(x).reshape(-1, 2); out = np.empty_like(x); pdf = np.empty(len(x), np.float32)
        fn(self.h, len(x), fp(x), fp(out), fp(pdf)); return out, pdf
    def sample(self, s): return self._run(lib().orc_hier2d_sample, s)
    def invert(self, p): return self._run(lib().orc_hier2d_invert, p)
    def eval(self, p):
        p = f32(p).reshape(-1, 2); pdf = np.empty(len(p), np.float32); lib().orc_hier2d_eval(self.h, len(p), fp(p), fp(pdf)); return pdf
    def storage(self):
        n = C.c_uint32(); size = lib().orc_hier2d_data(self.h, None, None, C.byref(n))
        data = np.empty(size, np.float32); table = np.empty((n.value, 3), np.uint32)
        lib().orc_hier2d_data(self.h, fp(data), up(table), C.byref(n)); return data, table


IDENTITY12 = [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]


class EnvMap:
    """EnvironmentMapEmitter (src/emitters/envmap.cpp) on an H x W x 3 image"""
    def __init__(self, rgb, scale=1.0, mis_compensation=False, to_world=IDENTITY12, to_local=IDENTITY12):
        rgb = f32(rgb); assert rgb.ndim == 3 and rgb.shape[2] == 3
        self.h = lib().orc_envmap_create(fp(rgb), rgb.shape[1], rgb.shape[0], scale, int(mis_compensation), fp(f32(to_world)), fp(f32(to_local)))
    def __del__(self):
        if getattr(self, "h", None): lib().orc_envmap_destroy(self.h)
    def set_bsphere(self, center, radius): lib().orc_envmap_set_bsphere(self.h, fp(f32(center)), radius)
    def eval(self, d):
        d = f32(d).reshape(-1, 3); out = np.empty_like(d); lib().orc_envmap_eval(self.h, len(d), fp(d), fp(out)); return out
    def pdf_direction(self, d):
        d = f32(d).reshape(-1, 3); out = np.empty(len(d), np.float32); lib().orc_envmap_pdf_direction(self.h, len(d), fp(d), fp(out)); return out
    def sample_direction(self, ref_p, sample):
        sample = f32(sample).reshape(-1, 2); n = len(sample); p = f32(np.broadcast_to(f32(ref_p).reshape(-1, 3), (n, 3)))
        d = np.empty((n, 3), np.float32); dist = np.empty(n, np.float32); pdf = np.empty(n, np.float32); w = np.empty((n, 3), np.float32)
        lib().orc_envmap_sample_direction(self.h, n, fp(p), fp(sample), fp(d), fp(dist), fp(pdf), fp(w)); return d, dist, pdf, w


class SceneData:
    """Flat scene description (numpy-backed) usable for BOTH the oracle and the
    product: `desc(cls)` re-emits it with the given ctypes struct classes."""

    def __init__(self):
        self.meshes = []     # dict(V, F, bsdf, emitter, flags)
        self.top_mesh_count = 0
        self.groups = []     # (first, count)
        self.instances = []  # (group, to_world12, to_object12)
        self.bsdfs = []      # (type, texture, rgb)
        self.textures = []   # HxWx3 float32
        self.texture_modes = []   # per texture: filter_type | wrap_mode (OrcTexture::mode), default 0 = bilinear + repeat
        self.texture_to_uv = []   # per texture: None or the six floats of the bitmap's `to_uv` (row-major 2 x 3)
        self.emitters = []   # dict(mesh, radiance, to_world12, normal, inv_area)
        self._keep = []

    def add_mesh(self, V, F, bsdf, emitter=-1, flags=3):
        self.meshes.append(dict(V=f32(V).reshape(-1, 8), F=np.ascontiguousarray(F, np.uint32).reshape(-1, 4),
                                bsdf=bsdf, emitter=emitter, flags=flags))
        return len(self.meshes) - 1

    def desc(self, ns=None):
        ns = ns or globals()
        M, G, I, B, TX, E, SD = (ns[k] for k in ("Mesh", "ShapeGroup", "Instance", "BSDF", "Texture", "Emitter", "SceneDesc"))
        keep = []
        meshes = (M * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            meshes[i].vertex_ptr = fp(m["V"]); meshes[i].index_ptr = up(m["F"])
            meshes[i].vertex_count = m["V"].shape[0]; meshes[i].face_count = m["F"].shape[0]
            meshes[i].bsdf = m["bsdf"]; meshes[i].emitter = m["emitter"]; meshes[i].flags = m["flags"]
        groups = (G * max(1, len(self.groups)))()
        for i, (a, b) in enumerate(self.groups):
            groups[i].first_mesh = a; groups[i].mesh_count = b
        insts = (I * max(1, len(self.instances)))()
        for i, (g, tw, to) in enumerate(self.instances):
            insts[i].group = g
            insts[i].to_world = (C.c_float * 12)(*[float(x) for x in tw])
            insts[i].to_object = (C.c_float * 12)(*[float(x) for x in to])
        bsdfs = (B * max(1, len(self.bsdfs)))()
        for i, entry in enumerate(self.bsdfs):
            # (type, texture, rgb) for diffuse, or (type, texture, rgb, extra) with extra = dict(flags=, reflectance2=, alpha_u=,
            # alpha_v=, eta=, eta_c=, k_c=, back=) for the other plugins
            t, tex, rgb = entry[0], entry[1], entry[2]
            x = entry[3] if len(entry) > 3 else {}
            bsdfs[i].type = t; bsdfs[i].texture = tex; bsdfs[i].reflectance = (C.c_float * 3)(*[float(v) for v in rgb])
            bsdfs[i].flags = int(x.get("flags", 0)); bsdfs[i].reflectance2 = (C.c_float * 3)(*[float(v) for v in x.get("reflectance2", (0, 0, 0))])
            bsdfs[i].alpha_u = float(x.get("alpha_u", 0.1)); bsdfs[i].alpha_v = float(x.get("alpha_v", 0.1)); bsdfs[i].eta = float(x.get("eta", 1.0))
            bsdfs[i].eta_c = (C.c_float * 3)(*[float(v) for v in x.get("eta_c", (0, 0, 0))]); bsdfs[i].k_c = (C.c_float * 3)(*[float(v) for v in x.get("k_c", (1, 1, 1))])
            bsdfs[i].back = int(x.get("back", -1))
        texs = (TX * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            texs[i].data = fp(t); texs[i].height = t.shape[0]; texs[i].width = t.shape[1]; texs[i].mode = self.texture_modes[i] if i < len(self.texture_modes) else 0
            uvm = self.texture_to_uv[i] if i < len(self.texture_to_uv) else None
            if uvm is not None:
                texs[i].to_uv = (C.c_float * 6)(*[float(x) for x in uvm])
        ems = (E * max(1, len(self.emitters)))()
        for i, e in enumerate(self.emitters):
            ems[i].type = int(e.get("type", 0)); ems[i].mesh = e["mesh"]
            ems[i].radiance = (C.c_float * 3)(*[float(x) for x in e["radiance"]])
            ems[i].to_world = (C.c_float * 12)(*[float(x) for x in e["to_world"]])
            ems[i].normal = (C.c_float * 3)(*[float(x) for x in e["normal"]])
            ems[i].inv_area = float(e["inv_area"])
            ems[i].to_local = (C.c_float * 12)(*[float(x) for x in e.get("to_local", [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0])])
            ems[i].sampling_weight = float(e.get("sampling_weight", 1.0))
            ems[i].radiance_texture = int(e.get("radiance_texture", 0))
        d = SD()
        d.meshes = meshes; d.mesh_count = len(self.meshes); d.top_mesh_count = self.top_mesh_count
        d.groups = groups; d.group_count = len(self.groups)
        d.instances = insts; d.instance_count = len(self.instances)
        d.bsdfs = bsdfs; d.bsdf_count = len(self.bsdfs)
        d.textures = texs; d.texture_count = len(self.textures)
        d.emitters = ems; d.emitter_count = len(self.emitters)
        keep += [meshes, groups, insts, bsdfs, texs, ems]
        self._keep.append(keep)
        return d


def rectangle(to_world):
    V = np.empty((4, 8), np.float32); F = np.empty((2, 4), np.uint32)
    n = np.empty(3, np.float32); ia = C.c_float()
    lib().orc_rectangle(fp(to_world.data), fp(V), up(F), fp(n), C.byref(ia))
    return V, F, n, ia.value


def cube(to_world):
    V = np.empty((24, 8), np.float32); F = np.empty((12, 4), np.uint32)
    lib().orc_cube(fp(to_world.data), fp(V), up(F))
    return V, F


def perspective_sensor(to_world, fov, fov_axis, near, far, width, height, crop=None, rfilter="gaussian", stddev=0.5):
    s = Sensor()
    cx, cy, cw, ch = crop if crop else (0, 0, width, height)
    lib().orc_perspective_sensor(fp(to_world.data), float(fov), fov_axis.encode(), near, far, width, height,
                                 cx, cy, cw, ch, 1 if rfilter == "gaussian" else 0, stddev, C.byref(s))
    return s


CBOX_WHITE = [0.885809, 0.698859, 0.666422]
CBOX_GREEN = [0.105421, 0.37798, 0.076425]
CBOX_RED = [0.570068, 0.0430135, 0.0443706]
CBOX_RADIANCE = [18.387, 13.9873, 6.75357]


def cornell_box(width=256, height=256, crop=None, rfilter="gaussian", white_texture=None):
    """Restates mi.cornell_box() (src/python/python/util.py:569-703)."""
    sd = SceneData()
    sd.bsdfs = [(0, -1, CBOX_WHITE), (0, -1, CBOX_GREEN), (0, -1, CBOX_RED)]
    if white_texture is not None:
        sd.textures.append(f32(white_texture))
        sd.bsdfs[0] = (0, 0, CBOX_WHITE)
    light_tf = T().translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19])
    V, F, n, ia = rectangle(light_tf)
    sd.add_mesh(V, F, 0, emitter=0)
    sd.emitters.append(dict(mesh=0, radiance=CBOX_RADIANCE, to_world=light_tf.col_major_3x4(), normal=n, inv_area=ia))
    for tf, b in [(T().translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90), 0),
                  (T().translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90), 0),
                  (T().translate([0.0, 0.0, -1.0]), 0),
                  (T().translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90), 1),
                  (T().translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90), 2)]:
        V, F, _, _ = rectangle(tf)
        sd.add_mesh(V, F, b)
    for tf in [T().translate([0.335, -0.7, 0.38]).rotate([0, 1, 0], -17).scale(0.3),
               T().translate([-0.33, -0.4, -0.28]).rotate([0, 1, 0], 18.25).scale([0.3, 0.61, 0.3])]:
        V, F = cube(tf)
        sd.add_mesh(V, F, 0)
    sd.top_mesh_count = len(sd.meshes)
    cam = T().look_at([0, 0, 3.9], [0, 0, 0], [0, 1, 0])
    sensor = perspective_sensor(cam, 39.3077, "smaller", 0.001, 100.0, width, height, crop, rfilter)
    return sd, sensor


def benchmark_spheres_scene(width=512, height=512, grid=10, n_u=100, n_v=50, flatten=False, textured=False, tex_res=256, rfilter="gaussian", materials=False):
    """The 1M-triangle benchmark scenes of SURVEY.md 8(d) (`mitsuba3_amd.scenes.instanced_spheres_scene`: Cornell box without its two
    boxes + grid x grid bumpy spheres, as instances of one shape group or flattened) lowered by the ORACLE's own code: its transform chain
    (orc_translate / orc_rotate / orc_scale / orc_matmul / orc_affine_inverse), its rectangle and mesh baking, its perspective sensor.  The
    tests that use this do not hand the product's baked vertex arrays, instance matrices or HarSensor to the oracle (`scene_from_product`
    does), so a defect in the product's host lowering of these scenes shows up as a parity failure.  The sphere's object-space vertex table
    and the checker bitmap are scene CONTENT (the asset the dict carries), taken from the same generator as the product's scene."""
    from mitsuba3_amd.scenes import bumpy_sphere, checker_texture
    sd = SceneData()
    sd.bsdfs = [(0, -1, CBOX_WHITE), (0, -1, CBOX_GREEN), (0, -1, CBOX_RED)]
    if textured:
        sd.textures.append(f32(checker_texture(tex_res))); sd.bsdfs[0] = (0, 0, CBOX_WHITE)
    n_ids = 3
    if materials:
        # the `materials=True` variant of the scene: `white` = roughplastic (src/bsdfs/roughplastic.cpp:151-195: Beckmann, visible normals, alpha 0.2, int_ior
        # "polypropylene" 1.49 over ext_ior "air" 1.000277 (src/bsdfs/ior.h), eta = int / ext in single precision, specular_reflectance 1), `green` = twosided GGX
        # roughconductor (roughconductor.cpp:155-205, twosided.cpp:70-110), `red` stays diffuse, fourth sphere material `glass` = dielectric with int_ior 1.5
        # (dielectric.cpp:160-185).  Written from the plugins' documented defaults, not read from the product's BSDF objects.
        assert flatten and not textured
        air = np.float32(1.000277)
        sd.bsdfs = [(3, -1, CBOX_WHITE, dict(flags=4, reflectance2=(1, 1, 1), alpha_u=0.2, alpha_v=0.2, eta=float(np.float32(1.49) / air))),
                    (2, -1, (1, 1, 1), dict(flags=1 | 2 | 4, alpha_u=0.15, alpha_v=0.15, eta_c=(0.2, 0.92, 1.1), k_c=(3.9, 2.45, 2.14))),
                    (0, -1, CBOX_RED),
                    (1, -1, (1, 1, 1), dict(reflectance2=(1, 1, 1), eta=float(np.float32(1.5) / air)))]
        n_ids = 4
    light_tf = T().translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19])
    V, F, n, ia = rectangle(light_tf)
    sd.add_mesh(V, F, 0, emitter=0)
    sd.emitters.append(dict(mesh=0, radiance=CBOX_RADIANCE, to_world=light_tf.col_major_3x4(), normal=n, inv_area=ia))
    for tf, b in [(T().translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90), 0), (T().translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90), 0),
                  (T().translate([0.0, 0.0, -1.0]), 0), (T().translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90), 1),
                  (T().translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90), 2)]:
        V, F, _, _ = rectangle(tf)
        sd.add_mesh(V, F, b)
    P, N, UV, Fi = bumpy_sphere(n_u, n_v)
    V0 = np.concatenate([P, N, UV], axis=1).astype(np.float32)
    F0 = np.concatenate([Fi, np.zeros((Fi.shape[0], 1), np.uint32)], axis=1).astype(np.uint32)
    tfs = []
    k = 0
    for gy in range(grid):
        for gx in range(grid):
            x = -0.8 + 1.6 * gx / max(grid - 1, 1); y = -0.85 + 1.5 * gy / max(grid - 1, 1)
            z = -0.5 + 0.9 * ((gx * 7 + gy * 3) % grid) / max(grid - 1, 1)
            tfs.append(T().translate([x, y, z]).rotate([0, 1, 0], 37.0 * k).scale(0.8 + 0.004 * k))
            k += 1
    if flatten:
        for k, tf in enumerate(tfs):
            V = V0.copy(); F = F0.copy()
            lib().orc_bake_mesh(fp(tf.data), fp(V), V.shape[0], up(F), F.shape[0])
            sd.add_mesh(V, F, k % n_ids)
        sd.top_mesh_count = len(sd.meshes)
    else:
        sd.top_mesh_count = len(sd.meshes)
        sd.add_mesh(V0, F0, 0)
        sd.groups = [(sd.top_mesh_count, 1)]
        sd.instances = [(0, tf.col_major_3x4(), tf.inverse().col_major_3x4()) for tf in tfs]
    cam = T().look_at([0, 0, 3.9], [0, 0, 0], [0, 1, 0])
    sensor = perspective_sensor(cam, 39.3077, "smaller", 0.001, 100.0, width, height, None, rfilter)
    return sd, sensor


class OracleScene:
    def __init__(self, scene_data):
        self.data = scene_data
        self._desc = scene_data.desc()
        self.handle = C.c_void_p(lib().orc_scene_create(C.byref(self._desc)))

    def __del__(self):
        try:
            if self.handle:
                lib().orc_scene_destroy(self.handle)
        except Exception:
            pass

    def ray_intersect(self, o, d, maxt, naive=False):
        o = f32(o); d = f32(d); maxt = f32(maxt); n = maxt.shape[0]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32); shape = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        lib().orc_ray_intersect(self.handle, n, fp(o), fp(d), fp(maxt), 1 if naive else 0,
                                fp(t), fp(u), fp(v), up(prim), up(shape), up(inst))
        return t, u, v, prim, shape, inst

    def ray_test(self, o, d, maxt, naive=False):
        o = f32(o); d = f32(d); maxt = f32(maxt); n = maxt.shape[0]
        hit = np.empty(n, np.uint8)
        lib().orc_ray_test(self.handle, n, fp(o), fp(d), fp(maxt), 1 if naive else 0,
                           hit.ctypes.data_as(C.POINTER(C.c_uint8)))
        return hit.astype(bool)

    def ray_intersect_masked(self, o, d, maxt, active, naive=False):
        o = f32(o); d = f32(d); maxt = f32(maxt); n = maxt.shape[0]
        a = np.ascontiguousarray(active, np.uint8)
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32); shape = np.empty(n, np.uint32); inst = np.empty(n, np.uint32)
        L = lib(); L.orc_ray_intersect_masked.restype = None
        L.orc_ray_intersect_masked.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_uint8), C.c_int, c_f32p, c_f32p, c_f32p, c_u32p, c_u32p, c_u32p]
        L.orc_ray_intersect_masked(self.handle, n, fp(o), fp(d), fp(maxt), a.ctypes.data_as(C.POINTER(C.c_uint8)), 1 if naive else 0, fp(t), fp(u), fp(v), up(prim), up(shape), up(inst))
        return t, u, v, prim, shape, inst

    def ray_test_masked(self, o, d, maxt, active, naive=False):
        o = f32(o); d = f32(d); maxt = f32(maxt); n = maxt.shape[0]
        a = np.ascontiguousarray(active, np.uint8); hit = np.empty(n, np.uint8)
        L = lib(); L.orc_ray_test_masked.restype = None
        L.orc_ray_test_masked.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8)]
        L.orc_ray_test_masked(self.handle, n, fp(o), fp(d), fp(maxt), a.ctypes.data_as(C.POINTER(C.c_uint8)), 1 if naive else 0, hit.ctypes.data_as(C.POINTER(C.c_uint8)))
        return hit.astype(bool)

    def surface_interaction_flags(self, o, d, t, u, v, prim, shape, inst, ray_flags=1, active=None):
        """compute_surface_interaction(ray, ray_flags, active) per lane: (33, n) rows p, n, sh_frame.n/.s/.t, wi, uv, t, dp_du, dp_dv, dn_du, dn_dv"""
        o = f32(o); d = f32(d); n = len(t)
        out = np.empty((33, n), np.float32); row = np.empty(33, np.float32)
        L = lib(); L.orc_surface_interaction_flags.restype = None
        L.orc_surface_interaction_flags.argtypes = [C.c_void_p, c_f32p, c_f32p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, c_f32p]
        for i in range(n):
            oo = f32(o[:, i]); dd = f32(d[:, i])
            L.orc_surface_interaction_flags(self.handle, fp(oo), fp(dd), float(t[i]), float(u[i]), float(v[i]), int(prim[i]), int(shape[i]), int(inst[i]) & 0xffffffff,
                                            int(ray_flags), 1 if (active is None or active[i]) else 0, fp(row))
            out[:, i] = row
        return out

    def bsdf_evaluate_ctx(self, bsdf, ctx, which, wi, uv, wo, active=None):
        """BSDF::eval (which 0) / pdf (1) / eval_pdf (2) with ctx = (mode, type_mask, component) and a mask: (value 3 x n, pdf n)"""
        wi = f32(wi); uv = f32(uv); wo = f32(wo); n = wo.shape[1]
        val = np.zeros((3, n), np.float32); pdf = np.zeros(n, np.float32); v3 = np.empty(3, np.float32); p1 = C.c_float()
        L = lib(); L.orc_bsdf_evaluate_ctx.restype = None
        L.orc_bsdf_evaluate_ctx.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, C.POINTER(C.c_float)]
        for i in range(n):
            a = f32(wi[:, i]); b = f32(uv[:, i]); c = f32(wo[:, i])
            L.orc_bsdf_evaluate_ctx(self.handle, bsdf, ctx[0], ctx[1] & 0xffffffff, ctx[2] & 0xffffffff, which, 1 if (active is None or active[i]) else 0, fp(a), fp(b), fp(c), fp(v3), C.byref(p1))
            val[:, i] = v3; pdf[i] = p1.value
        return val, pdf

    def bsdf_sample_ctx(self, bsdf, ctx, wi, uv, s1, s2, active=None):
        """BSDF::sample with ctx and a mask: dict(wo 3 x n, pdf, weight 3 x n, eta, sampled_type, sampled_component)"""
        wi = f32(wi); uv = f32(uv); s1 = f32(s1); s2 = f32(s2); n = s2.shape[1]
        wo = np.zeros((3, n), np.float32); w = np.zeros((3, n), np.float32); pdf = np.zeros(n, np.float32); eta = np.zeros(n, np.float32)
        st = np.zeros(n, np.uint32); sc = np.zeros(n, np.uint32)
        a3 = np.empty(3, np.float32); w3 = np.empty(3, np.float32); p1 = C.c_float(); e1 = C.c_float(); t1 = C.c_uint32(); c1 = C.c_uint32()
        L = lib(); L.orc_bsdf_sample_ctx.restype = None
        L.orc_bsdf_sample_ctx.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, C.POINTER(C.c_float), c_f32p,
                                          C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        for i in range(n):
            a = f32(wi[:, i]); b = f32(uv[:, i]); c = f32(s2[:, i])
            L.orc_bsdf_sample_ctx(self.handle, bsdf, ctx[0], ctx[1] & 0xffffffff, ctx[2] & 0xffffffff, 1 if (active is None or active[i]) else 0, fp(a), fp(b), float(s1[i]), fp(c),
                                  fp(a3), C.byref(p1), fp(w3), C.byref(e1), C.byref(t1), C.byref(c1))
            wo[:, i] = a3; w[:, i] = w3; pdf[i] = p1.value; eta[i] = e1.value; st[i] = t1.value; sc[i] = c1.value
        return dict(wo=wo, pdf=pdf, weight=w, eta=eta, sampled_type=st, sampled_component=sc)

    def _render(self, fn, sensor, seed, spp, max_depth, rr_depth, lanes, threads):
        film = np.zeros((sensor.crop_height, sensor.crop_width, 4), np.float32)
        st = Stats()
        lb, le = lanes if lanes else (0, 0)
        rc = fn(self.handle, C.byref(sensor), seed, spp, max_depth, rr_depth, lb, le, fp(film), C.byref(st), threads)
        assert rc == 0
        return film, st

    def render_path(self, sensor, seed=0, spp=4, max_depth=8, rr_depth=5, lanes=None, threads=0, raw=False):
        film, st = self._render(lib().orc_render_path, sensor, seed, spp, max_depth, rr_depth, lanes, threads)
        return (film if raw else develop(film)), st

    def render_path_passes(self, sensor, seed=0, spp=4, spp_per_pass=2, max_depth=8, rr_depth=5, lanes=None, threads=0, raw=False):
        """multi-pass JIT render: `lanes` index the per-pass wavefront of W*H*spp_per_pass lanes"""
        film = np.zeros((sensor.crop_height, sensor.crop_width, 4), np.float32)
        st = Stats(); lb, le = lanes if lanes else (0, 0)
        rc = lib().orc_render_path_passes(self.handle, C.byref(sensor), seed, spp, spp_per_pass, max_depth, rr_depth, lb, le, fp(film), C.byref(st), threads)
        assert rc == 0, rc
        return (film if raw else develop(film)), st

    def render_path_scalar(self, sensor, seed=0, spp=4, max_depth=8, rr_depth=5, n_threads=1, raw=False):
        """`scalar_rgb` driver (spiral blocks, Morton order, per-pixel reseed, discretised filter): BASELINE config 1"""
        film = np.zeros((sensor.crop_height, sensor.crop_width, 4), np.float32)
        st = Stats(); bs = C.c_uint32()
        rc = lib().orc_render_path_scalar(self.handle, C.byref(sensor), seed, spp, max_depth, rr_depth, n_threads, fp(film), C.byref(st), C.cast(C.byref(bs), c_u32p))
        assert rc == 0
        return (film if raw else develop(film)), st, bs.value

    def render_prb(self, sensor, seed=0, spp=4, max_depth=6, rr_depth=5, lanes=None, threads=0, raw=False):
        film, st = self._render(lib().orc_render_prb, sensor, seed, spp, max_depth, rr_depth, lanes, threads)
        return (film if raw else develop(film)), st

    def render_prb_backward(self, sensor, grad_in, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        grad_in = f32(grad_in)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        st = Stats()
        rc = lib().orc_render_prb_backward(self.handle, C.byref(sensor), fp(grad_in), seed, spp, max_depth, rr_depth,
                                           fp(g_refl), ptrs, C.byref(st), threads)
        assert rc == 0
        return g_refl, g_tex, st

    def render_prb_backward_emitters(self, sensor, grad_in, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        """as render_prb_backward, plus the gradient w.r.t. emitter radiances (emitter_count x 3)"""
        grad_in = f32(grad_in)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32); g_emit = np.zeros((max(1, len(self.data.emitters)), 3), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        st = Stats()
        rc = lib().orc_render_prb_backward_ex(self.handle, C.byref(sensor), fp(grad_in), seed, spp, max_depth, rr_depth,
                                              fp(g_refl), ptrs, fp(g_emit), C.byref(st), threads)
        assert rc == 0
        return g_refl, g_tex, g_emit[:len(self.data.emitters)], st

    def render_prb_backward_bsdf_params(self, sensor, grad_in, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        """gradients w.r.t. alpha_u, alpha_v, eta, k, colour slot 1 of every BSDF record: array (bsdf_count, 5, 3) of per-channel contributions"""
        grad_in = f32(grad_in)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32); g_x = np.zeros((len(self.data.bsdfs), 15), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        st = Stats()
        L = lib(); L.orc_render_prb_backward_bsdf_params.restype = C.c_int
        L.orc_render_prb_backward_bsdf_params.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, c_f32p, C.POINTER(c_f32p), c_f32p,
                                                          C.POINTER(Stats), C.c_int]
        rc = L.orc_render_prb_backward_bsdf_params(self.handle, C.byref(sensor), fp(grad_in), seed, spp, max_depth, rr_depth, fp(g_refl), ptrs, fp(g_x), C.byref(st), threads)
        assert rc == 0
        return g_x.reshape(-1, 5, 3), g_refl

    def render_prb_forward(self, sensor, t_refl, t_tex=(), t_emit=None, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0, raw=False):
        """RBIntegrator.render_forward: gradient image (H x W x 3) for the parameter tangents t_refl (bsdf_count x 3), t_tex (one array per bitmap),
        t_emit (emitter_count x 3 or None)"""
        t_refl = f32(t_refl); t_tex = [f32(t) for t in t_tex]
        ptrs = (c_f32p * max(1, len(t_tex)))(*[fp(t) for t in t_tex])
        te = None if t_emit is None else f32(t_emit)
        film = np.zeros((sensor.crop_height, sensor.crop_width, 4), np.float32)
        L = lib(); L.orc_render_prb_forward.restype = C.c_int
        L.orc_render_prb_forward.argtypes = [C.c_void_p, C.POINTER(Sensor), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, c_f32p, C.POINTER(c_f32p), c_f32p, c_f32p, C.c_int]
        rc = L.orc_render_prb_forward(self.handle, C.byref(sensor), seed, spp, max_depth, rr_depth, fp(t_refl), ptrs, fp(te) if te is not None else None, fp(film), threads)
        assert rc == 0
        return film if raw else develop(film)

    def integrator_sample(self, o, d, maxt, seed=0, lane_offset=0, state=None, max_depth=8, rr_depth=5, prb=False, threads=0, active=None):
        """SamplingIntegrator::sample over n rays (3 x n origins / directions): (rgb 3 x n, valid n uint8, state_out n uint64); active: the Mask argument"""
        o = f32(o); d = f32(d); maxt = f32(maxt); n = maxt.shape[0]
        rgb = np.empty((3, n), np.float32); valid = np.empty(n, np.uint8); so = np.empty(n, np.uint64)
        st = None if state is None else np.ascontiguousarray(state, np.uint64)
        am = None if active is None else np.ascontiguousarray(active, np.uint8)
        L = lib(); L.orc_integrator_sample_masked.restype = C.c_int
        u64p = C.POINTER(C.c_uint64)
        L.orc_integrator_sample_masked.argtypes = [C.c_void_p, C.c_int, C.c_uint32, c_f32p, c_f32p, c_f32p, C.c_uint32, C.c_uint32, u64p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, c_f32p,
                                                   C.POINTER(C.c_uint8), u64p, C.c_int]
        rc = L.orc_integrator_sample_masked(self.handle, 1 if prb else 0, n, fp(o), fp(d), fp(maxt), seed, lane_offset, st.ctypes.data_as(u64p) if st is not None else None,
                                            am.ctypes.data_as(C.POINTER(C.c_uint8)) if am is not None else None,
                                            max_depth, rr_depth, fp(rgb), valid.ctypes.data_as(C.POINTER(C.c_uint8)), so.ctypes.data_as(u64p), threads)
        assert rc == 0
        return rgb, valid, so

    def render_prb_backward_lanes(self, sensor, grad_in, weight_film, lanes, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        """one rank's share of a multi-GPU render_backward: lanes [lo, hi) against the all-reduced weight film (H x W x 4).
        Returns (g_refl, g_tex, g_emit, stats); sums over disjoint lane bands equal render_prb_backward_emitters of the whole frame."""
        grad_in = f32(grad_in); weight_film = f32(weight_film)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32); g_emit = np.zeros((max(1, len(self.data.emitters)), 3), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        st = Stats()
        L = lib(); L.orc_render_prb_backward_lanes.restype = C.c_int
        L.orc_render_prb_backward_lanes.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64,
                                                    c_f32p, C.POINTER(c_f32p), c_f32p, C.POINTER(Stats), C.c_int]
        rc = L.orc_render_prb_backward_lanes(self.handle, C.byref(sensor), fp(grad_in), fp(weight_film), seed, spp, max_depth, rr_depth, lanes[0], lanes[1],
                                             fp(g_refl), ptrs, fp(g_emit), C.byref(st), threads)
        assert rc == 0
        return g_refl, g_tex, g_emit[:len(self.data.emitters)], st

    def render_prb_backward_shape(self, sensor, grad_in, meshes, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        """as render_prb_backward, plus {mesh index: (vertex_count, 3) float64 gradient w.r.t. its vertex positions}"""
        grad_in = f32(grad_in); nm = len(self.data.meshes)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        mask = np.zeros(nm, np.uint8); mask[list(meshes)] = 1
        g_pos = {m: np.zeros((self.data.meshes[m]["V"].shape[0], 3), np.float64) for m in meshes}
        dp = C.POINTER(C.c_double)
        pp = (dp * nm)(*[g_pos[m].ctypes.data_as(dp) if m in g_pos else dp() for m in range(nm)])
        st = Stats()
        L = lib(); L.orc_render_prb_backward_shape.restype = C.c_int
        L.orc_render_prb_backward_shape.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, c_f32p,
                                                    C.POINTER(c_f32p), C.c_void_p, C.POINTER(dp), C.POINTER(Stats), C.c_int]
        rc = L.orc_render_prb_backward_shape(self.handle, C.byref(sensor), fp(grad_in), seed, spp, max_depth, rr_depth, fp(g_refl), ptrs,
                                             mask.ctypes.data, pp, C.byref(st), threads)
        if rc != 0:
            raise RuntimeError("orc_render_prb_backward_shape: rc = %d" % rc)
        return g_pos, g_refl, g_tex, st

    def render_prb_backward_instances(self, sensor, grad_in, instances=None, seed=0, spp=4, max_depth=6, rr_depth=5, threads=0):
        """as render_prb_backward, plus (instance_count, 3, 4) float64: d loss / d to_world (row r, column c; the 4th row of the 4x4 is constant)"""
        grad_in = f32(grad_in); ni = len(self.data.instances)
        g_refl = np.zeros((len(self.data.bsdfs), 3), np.float32)
        g_tex = [np.zeros_like(t) for t in self.data.textures]
        ptrs = (c_f32p * max(1, len(g_tex)))(*[fp(g) for g in g_tex])
        mask = np.zeros(max(1, ni), np.uint8); mask[list(range(ni)) if instances is None else list(instances)] = 1
        g = np.zeros((max(1, ni), 12), np.float64)
        st = Stats()
        L = lib(); L.orc_render_prb_backward_instances.restype = C.c_int
        L.orc_render_prb_backward_instances.argtypes = [C.c_void_p, C.POINTER(Sensor), c_f32p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, c_f32p,
                                                        C.POINTER(c_f32p), C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
        rc = L.orc_render_prb_backward_instances(self.handle, C.byref(sensor), fp(grad_in), seed, spp, max_depth, rr_depth, fp(g_refl), ptrs,
                                                 mask.ctypes.data, g.ctypes.data, C.byref(st), threads)
        if rc != 0:
            raise RuntimeError("orc_render_prb_backward_instances: rc = %d" % rc)
        return g[:ni].reshape(ni, 4, 3).transpose(0, 2, 1).copy(), g_refl, g_tex, st       # column-major 3x4 -> [row][col]

    def set_instance_to_world(self, inst, m4):
        """new to_world (4x4, affine) of an instance + rebuild of the instance-level acceleration structure"""
        m = np.asarray(m4, np.float64).reshape(4, 4); inv = np.linalg.inv(m)
        tw = f32(m[:3, :].T.reshape(-1)); to = f32(inv[:3, :].T.reshape(-1))
        L = lib(); L.orc_scene_set_instance_transform.restype = None; L.orc_scene_set_instance_transform.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_f32p]
        L.orc_scene_set_instance_transform(self.handle, inst, fp(tw), fp(to))

    def set_vertex_positions(self, mesh, positions):
        p = f32(positions).reshape(-1, 3)
        assert p.shape[0] == self.data.meshes[mesh]["V"].shape[0]
        L = lib(); L.orc_scene_set_vertex_positions.restype = None; L.orc_scene_set_vertex_positions.argtypes = [C.c_void_p, C.c_uint32, c_f32p]
        L.orc_scene_set_vertex_positions(self.handle, mesh, fp(p))
        self.data.meshes[mesh]["V"][:, :3] = p

    def set_alpha_only(self, on):
        L = lib(); L.orc_scene_set_alpha_only.argtypes = [C.c_void_p, C.c_int]; L.orc_scene_set_alpha_only.restype = None
        L.orc_scene_set_alpha_only(self.handle, 1 if on else 0)

    def set_hide_emitters(self, hide):
        lib().orc_scene_set_hide_emitters(self.handle, 1 if hide else 0)

    def set_emitter_radiance(self, emitter, rgb):
        lib().orc_scene_set_emitter_radiance(self.handle, emitter, fp(f32(rgb)))

    def set_reflectance(self, bsdf, rgb):
        lib().orc_scene_set_reflectance(self.handle, bsdf, fp(f32(rgb)))

    def sample_emitter(self, sample):
        """Scene::sample_emitter (scene.cpp:248-271): (index, weight, reused sample) arrays"""
        sample = f32(sample).reshape(-1); n = sample.size
        index = np.zeros(n, np.uint32); weight = np.zeros(n, np.float32); reused = np.zeros(n, np.float32)
        L = lib(); L.orc_scene_sample_emitter.restype = None
        L.orc_scene_sample_emitter.argtypes = [C.c_void_p, C.c_uint32, c_f32p, c_u32p, c_f32p, c_f32p]
        L.orc_scene_sample_emitter(self.handle, n, fp(sample), index.ctypes.data_as(c_u32p), fp(weight), fp(reused))
        return index, weight, reused

    def pdf_emitter(self, index):
        index = np.ascontiguousarray(index, np.uint32).reshape(-1); pdf = np.zeros(index.size, np.float32)
        L = lib(); L.orc_scene_pdf_emitter.restype = None; L.orc_scene_pdf_emitter.argtypes = [C.c_void_p, C.c_uint32, c_u32p, c_f32p]
        L.orc_scene_pdf_emitter(self.handle, index.size, index.ctypes.data_as(c_u32p), fp(pdf))
        return pdf

    def set_emitter_weights(self, weights):
        w = f32(weights).reshape(-1)
        L = lib(); L.orc_scene_set_emitter_weights.restype = C.c_int; L.orc_scene_set_emitter_weights.argtypes = [C.c_void_p, c_f32p, C.c_uint32]
        rc = L.orc_scene_set_emitter_weights(self.handle, fp(w), w.size)
        if rc:
            raise RuntimeError("DiscreteDistribution: invalid emitter weights (rc %d)" % rc)

    def set_texture_to_uv(self, idx, rows):
        m = f32(rows).reshape(-1)
        L = lib(); L.orc_scene_set_texture_to_uv.restype = None; L.orc_scene_set_texture_to_uv.argtypes = [C.c_void_p, C.c_uint32, c_f32p]
        L.orc_scene_set_texture_to_uv(self.handle, idx, fp(m))

    def set_texture(self, idx, data):
        lib().orc_scene_set_texture(self.handle, idx, fp(f32(data)))


def develop(film):
    h, w, _ = film.shape
    img = np.empty((h, w, 3), np.float32)
    lib().orc_film_develop(fp(f32(film)), w, h, fp(img))
    return img


def render_weights(sensor, seed, spp, lanes=None, threads=0):
    """weight-only splat (common.py:716-746) of lanes [lo, hi) (None = all): raw film H x W x 4 with the accumulated weights in channel 3"""
    film = np.zeros((sensor.crop_height, sensor.crop_width, 4), np.float32)
    L = lib(); L.orc_render_weights.restype = C.c_int
    L.orc_render_weights.argtypes = [C.POINTER(Sensor), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, c_f32p, C.c_int]
    lo, hi = lanes if lanes else (0, 0)
    assert L.orc_render_weights(C.byref(sensor), seed, spp, lo, hi, fp(film), threads) == 0
    return film


def scene_from_product(scene):
    """Feed the product's flat scene arrays (mitsuba3_amd.Scene) to the oracle unchanged."""
    sd = SceneData()
    if hasattr(scene, "sync_host"):
        scene.sync_host()          # values params.update() pushed device-to-device -> the numpy mirrors read below
    for m in scene.meshes:
        sd.add_mesh(m["V"], m["F"], m["bsdf"], m["emitter"], m["flags"])
    sd.top_mesh_count = scene.top_mesh_count
    sd.groups = list(scene.groups); sd.instances = list(scene.instances)
    types = {"diffuse": 0, "dielectric": 1, "roughconductor": 2, "roughplastic": 3, "conductor": 4, "plastic": 5}
    sd.bsdfs = [(types[b.kind], b.tex_index if b.texture is not None else -1, b.value,
                 dict(flags=b.flags, reflectance2=b.value2, alpha_u=b.alpha_u, alpha_v=b.alpha_v, eta=b.eta, eta_c=b.eta_c, k_c=b.k_c,
                      back=b.back.index if b.back is not None else -1)) for b in scene.bsdf_objs]
    sd.textures = list(scene.textures); sd.texture_modes = list(getattr(scene, 'texture_modes', [])); sd.emitters = list(scene.emitters)
    sd.texture_to_uv = list(getattr(scene, 'texture_to_uv', []))
    s = Sensor()
    if scene.sensors():          # (a scene without a sensor still answers ray and emitter queries)
        C.memmove(C.byref(s), C.byref(scene.sensors()[0].har), C.sizeof(s))
    return OracleScene(sd), s
