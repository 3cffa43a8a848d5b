"""Host-side mirror of the Mitsuba 3 Python surface for the `hip_ad_rgb` variant.

Names, argument meaning and error behaviour follow the reference
(`mi.load_dict`, `mi.cornell_box`, `mi.render`, `mi.traverse`,
`Scene.ray_intersect_preliminary`, `Integrator.render/render_backward`,
`Sampler.seed/next_1d/next_2d`, `BSDF.eval/pdf/sample`); all computation is
done by the HIP kernels behind the C ABI (mitsuba3_amd/_capi.py).  PyTorch is
used for device memory, streams and torch.distributed only.
"""
import ctypes as C
import math
import os
import re

import numpy as np

from . import _capi
from ._capi import lib, check, HarError

VARIANT = "hip_ad_rgb"
SCALAR_VARIANT = "scalar_rgb"        # BASELINE config 1: the reference's CPU plumbing path (har_render_scalar), forward `path` renders only
_variant = None


def set_variant(name):
    """mi.set_variant: `hip_ad_rgb` (the HIP hot path; needs a GPU, no fallback) or `scalar_rgb` (config 1: the scalar branch of
    SamplingIntegrator::render on the host, forward `path` only -- an explicit choice, never a fallback of the former)."""
    global _variant
    if name not in (VARIANT, SCALAR_VARIANT):
        raise ImportError("Requested an unsupported variant \"%s\". The following variants are available: %s, %s." % (name, VARIANT, SCALAR_VARIANT))
    _variant = name


def variant():
    return _variant


def variants():
    return [VARIANT, SCALAR_VARIANT]


def _render_scalar(scene, integrator, sensor, seed, spp, threads=0, block_size=0):
    """SamplingIntegrator::render, non-JIT branch (integrator.cpp:190-274) through har_render_scalar: developed image as a numpy array"""
    if integrator.type != 'path':
        raise RuntimeError("scalar_rgb: only the `path` integrator is part of the config-1 plumbing path")
    if integrator.hide_emitters or sensor.film().alpha or integrator.samples_per_pass is not None:
        raise RuntimeError("scalar_rgb: hide_emitters, rgba films and samples_per_pass are not part of the config-1 plumbing path")
    if spp:
        sensor.sampler().set_sample_count(spp)
    spp = sensor.sampler().sample_count()
    w, h = sensor.film().crop_size()
    film = np.zeros((h, w, 4), np.float32)
    d = scene.desc()
    used = C.c_uint32(0)
    # the scalar driver passes `seed` through Sampler::seed, which adds the sampler's own `seed` property (sampler.cpp:129-131)
    check(lib().har_render_scalar(C.byref(d), C.byref(sensor.har), int(seed) & 0xffffffff, spp, integrator.max_depth, integrator.rr_depth, block_size, threads,
                                   film.ctypes.data_as(C.c_void_p), C.byref(used)))
    wgt = film[..., 3:4]
    rgb = film[..., :3]
    rows = _COLOUR_ROWS[sensor.film().colour]           # `luminance` / `xyz` films: the colour transform of HDRFilm::develop on the weighted sums, then / W
    if rows is not None:
        rgb = rgb @ np.asarray(rows, np.float32).T
    return rgb / np.where(wgt == 0, 1.0, wgt).astype(np.float32)


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_capi.f32p)


def _up(a):
    return a.ctypes.data_as(_capi.u32p)


def _torch():
    import torch
    return torch


def _device():
    torch = _torch()
    if not torch.cuda.is_available():
        raise HarError("hip_ad_rgb requires a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    _capi.install_torch_allocator()         # once: the library allocates through PyTorch's caching allocator (har_set_allocator)
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------------------
#  ScalarTransform4f (include/mitsuba/core/transform.h) -- evaluated in C++
# ---------------------------------------------------------------------------

class ScalarTransform4f:
    def __init__(self, data=None):
        if data is None:
            data = np.concatenate([np.eye(4, dtype=np.float32).ravel()] * 2)
        self.data = _f32(data).reshape(32)

    def _chain(self, other):
        out = np.empty(32, np.float32)
        check(lib().har_transform_mul(_fp(self.data), _fp(other), _fp(out)))
        return ScalarTransform4f(out)

    def translate(self, v):
        out = np.empty(32, np.float32); check(lib().har_transform_translate(_fp(_f32(v)), _fp(out))); return self._chain(out)

    def scale(self, v):
        v = _f32([v, v, v]) if np.isscalar(v) else _f32(v)
        out = np.empty(32, np.float32); check(lib().har_transform_scale(_fp(v), _fp(out))); return self._chain(out)

    def rotate(self, axis, angle):
        out = np.empty(32, np.float32); check(lib().har_transform_rotate(_fp(_f32(axis)), float(angle), _fp(out))); return self._chain(out)

    def look_at(self, origin, target, up):
        out = np.empty(32, np.float32)
        check(lib().har_transform_look_at(_fp(_f32(origin)), _fp(_f32(target)), _fp(_f32(up)), _fp(out))); return self._chain(out)

    def inverse(self):
        out = np.empty(32, np.float32); check(lib().har_transform_inverse(_fp(self.data), _fp(out))); return ScalarTransform4f(out)

    def __matmul__(self, other):
        return self._chain(other.data)

    @property
    def matrix(self):
        return self.data[:16].reshape(4, 4)

    def col_major_3x4(self):
        return np.ascontiguousarray(self.matrix[:3, :].T).ravel().astype(np.float32)

    def has_scale(self):
        m = self.matrix[:3, :3].astype(np.float64)
        return not np.allclose(m @ m.T, np.eye(3), atol=1e-3)


Transform4f = ScalarTransform4f


class ScalarTransform3f:
    """ScalarAffineTransform3f (include/mitsuba/core/transform.h): 2-D affine maps, the type of BitmapTexture's `to_uv`.  Chained like the reference's Python API:
    T().translate(v).rotate(deg).scale(v) = translate * rotate * scale (the last call acts first on a point)."""

    def __init__(self, matrix=None):
        self.matrix = np.eye(3, dtype=np.float32) if matrix is None else np.asarray(matrix, np.float32).reshape(3, 3).copy()

    def _chain(self, m):
        # Transform::operator*, affine branch (transform.h:364-400): sums of fmadd in float32 -- float64 accumulation rounded once is the same value for these 2-term sums
        # except in the last bit; `to_uv` is consumed as six float32 numbers, so the stored matrix IS the parameter
        return ScalarTransform3f((self.matrix.astype(np.float64) @ np.asarray(m, np.float64)).astype(np.float32))

    def translate(self, v):
        m = np.eye(3, dtype=np.float32); m[0, 2] = v[0]; m[1, 2] = v[1]; return self._chain(m)

    def scale(self, v):
        v = [v, v] if np.isscalar(v) else v
        m = np.eye(3, dtype=np.float32); m[0, 0] = v[0]; m[1, 1] = v[1]; return self._chain(m)

    def rotate(self, angle):
        a = math.radians(float(angle)); c, sn = math.cos(a), math.sin(a)
        m = np.eye(3, dtype=np.float32); m[0, 0] = c; m[0, 1] = -sn; m[1, 0] = sn; m[1, 1] = c; return self._chain(m)

    def inverse(self):
        return ScalarTransform3f(np.linalg.inv(self.matrix.astype(np.float64)).astype(np.float32))

    def __matmul__(self, other):
        return self._chain(other.matrix)

    def rows_2x3(self):
        return [float(x) for x in self.matrix[:2, :].reshape(-1)]


Transform3f = ScalarTransform3f


def cornell_box():
    """mi.cornell_box() (src/python/python/util.py:569-703)."""
    T = ScalarTransform4f
    return {
        'type': 'scene',
        'integrator': {'type': 'path', 'max_depth': 8},
        'sensor': {
            'type': 'perspective', 'fov_axis': 'smaller', 'near_clip': 0.001, 'far_clip': 100.0,
            'focus_distance': 1000, 'fov': 39.3077,
            'to_world': T().look_at(origin=[0, 0, 3.90], target=[0, 0, 0], up=[0, 1, 0]),
            'sampler': {'type': 'independent', 'sample_count': 64},
            'film': {'type': 'hdrfilm', 'width': 256, 'height': 256, 'rfilter': {'type': 'gaussian'},
                     'pixel_format': 'rgb', 'component_format': 'float32'},
        },
        'white': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.885809, 0.698859, 0.666422]}},
        'green': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.105421, 0.37798, 0.076425]}},
        'red': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.570068, 0.0430135, 0.0443706]}},
        'light': {'type': 'rectangle',
                  'to_world': T().translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19]),
                  'bsdf': {'type': 'ref', 'id': 'white'},
                  'emitter': {'type': 'area', 'radiance': {'type': 'rgb', 'value': [18.387, 13.9873, 6.75357]}}},
        'floor': {'type': 'rectangle', 'to_world': T().translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'ceiling': {'type': 'rectangle', 'to_world': T().translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'back': {'type': 'rectangle', 'to_world': T().translate([0.0, 0.0, -1.0]), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'green-wall': {'type': 'rectangle', 'to_world': T().translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90), 'bsdf': {'type': 'ref', 'id': 'green'}},
        'red-wall': {'type': 'rectangle', 'to_world': T().translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90), 'bsdf': {'type': 'ref', 'id': 'red'}},
        'small-box': {'type': 'cube', 'to_world': T().translate([0.335, -0.7, 0.38]).rotate([0, 1, 0], -17).scale(0.3), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'large-box': {'type': 'cube', 'to_world': T().translate([-0.33, -0.4, -0.28]).rotate([0, 1, 0], 18.25).scale([0.3, 0.61, 0.3]), 'bsdf': {'type': 'ref', 'id': 'white'}},
    }


# ---------------------------------------------------------------------------
#  Wavefront record types (SoA torch tensors on the GPU)
# ---------------------------------------------------------------------------

class Ray3f:
    """Ray3f wavefront: o, d are float32 tensors [3, n]; maxt [n] (default Largest)."""

    def __init__(self, o, d, maxt=None):
        torch = _torch(); dev = _device()
        self.o = torch.as_tensor(o, dtype=torch.float32, device=dev).reshape(3, -1).contiguous()
        self.d = torch.as_tensor(d, dtype=torch.float32, device=dev).reshape(3, -1).contiguous()
        n = self.o.shape[1]
        if maxt is None:
            maxt = torch.full((n,), 3.402823466e+38, dtype=torch.float32, device=dev)
        self.maxt = torch.as_tensor(maxt, dtype=torch.float32, device=dev).reshape(-1).contiguous()
        if self.maxt.numel() == 1 and n > 1:
            self.maxt = self.maxt.expand(n).contiguous()

    def __len__(self):
        return self.o.shape[1]


class PreliminaryIntersection3f:
    """include/mitsuba/render/interaction.h:717-836"""

    def __init__(self, scene, t, u, v, prim_index, shape_index, inst_index):
        self.scene = scene
        self.t = t; self.prim_uv = (u, v); self.prim_index = prim_index
        self.shape_index = shape_index; self.instance = inst_index

    def is_valid(self):
        return ~_torch().isinf(self.t)

    def compute_surface_interaction(self, ray, ray_flags=None, active=True):
        """PreliminaryIntersection::compute_surface_interaction(ray, ray_flags = RayFlags::Default, active) (interaction.h:804-829)"""
        return self.scene._compute_si(ray, self, RayFlags.Default if ray_flags is None else ray_flags, active)


class SurfaceInteraction3f:
    """rows of har_compute_surface_interaction's output (include/hip_ad_rgb.h HAR_SI_ROWS; interaction.h:345-420)"""

    def __init__(self, raw):
        self.p = raw[0:3]; self.n = raw[3:6]
        self.sh_frame = type("Frame3f", (), dict(n=raw[6:9], s=raw[9:12], t=raw[12:15]))()
        self.wi = raw[15:18]; self.uv = raw[18:20]; self.t = raw[20]
        self.dp_du = raw[21:24]; self.dp_dv = raw[24:27]; self.dn_du = raw[27:30]; self.dn_dv = raw[30:33]

    def is_valid(self):
        return ~_torch().isinf(self.t)


class RayFlags:
    """include/mitsuba/render/interaction.h:19-87"""
    Minimal = 0x0; Shading = 0x1; NormalPartials = 0x2; Default = 0x1; All = 0x1; FollowShape = 0x4; DetachShape = 0x8


class TransportMode:
    """include/mitsuba/render/fwd.h / bsdf.h: Radiance = 0, Importance = 1"""
    Radiance = 0; Importance = 1


class BSDFFlags:
    """include/mitsuba/render/bsdf.h:31-125 (the lobe types and their unions)"""
    Empty = 0x0; Null = 0x1; DiffuseReflection = 0x2; DiffuseTransmission = 0x4; GlossyReflection = 0x8; GlossyTransmission = 0x10
    DeltaReflection = 0x20; DeltaTransmission = 0x40; Delta1DReflection = 0x80; Delta1DTransmission = 0x100
    Reflection = 0x2 | 0x20 | 0x80 | 0x8; Transmission = 0x4 | 0x40 | 0x100 | 0x10 | 0x1
    Diffuse = 0x2 | 0x4; Glossy = 0x8 | 0x10; Smooth = 0x2 | 0x4 | 0x8 | 0x10; Delta = 0x1 | 0x20 | 0x40; Delta1D = 0x80 | 0x100; All = 0x1ff


class BSDFContext:
    """include/mitsuba/render/bsdf.h:140-186: transport mode, lobe type mask, component index; every BSDF call honours all three (har_bsdf_*)"""

    def __init__(self, mode=TransportMode.Radiance, type_mask=0x1ff, component=0xffffffff):
        self.mode = int(mode); self.type_mask = int(type_mask) & 0xffffffff; self.component = int(component) & 0xffffffff

    def reverse(self):
        self.mode = 1 - self.mode

    def is_enabled(self, type_, component_=0):
        t = int(type_)
        return (self.type_mask == 0xffffffff or (self.type_mask & t) == t) and (self.component == 0xffffffff or self.component == component_)

    def _c(self):
        from ._capi import HarBSDFContext
        if self.mode not in (0, 1):
            raise RuntimeError("BSDFContext.mode must be TransportMode.Radiance or TransportMode.Importance")
        return C.byref(HarBSDFContext(self.mode, self.type_mask, self.component))


def _mask(active, n):
    """the reference's `Mask active` argument -> device uint8[n] (None when every lane is active: the C ABI's NULL)"""
    if active is True or active is None:
        return None
    torch = _torch()
    if active is False:
        return torch.zeros(n, dtype=torch.uint8, device=_device())
    a = torch.as_tensor(active, device=_device()).reshape(-1)
    a = (a != 0).to(torch.uint8)
    if a.numel() == 1 and n != 1:
        a = a.expand(n)
    if a.numel() != n:
        raise RuntimeError("active: mask of %d entries for a wavefront of %d lanes" % (a.numel(), n))
    return a.contiguous()


def _ctx(ctx):
    if ctx is None:
        return None
    if not isinstance(ctx, BSDFContext):
        raise RuntimeError("ctx must be a BSDFContext (or None for the default context)")
    return ctx._c()


# ---------------------------------------------------------------------------
#  Plugins
# ---------------------------------------------------------------------------

class Mesh:
    """Triangle mesh in the packed layout (mesh_utils.h:19-34)."""

    def __init__(self, name="mesh"):
        self.name = name
        self.V = np.zeros((0, 8), np.float32); self.F = np.zeros((0, 4), np.uint32); self.flags = 0
        self.bsdf = None; self.emitter = None

    def from_fields(self, faces, positions, normals=None, texcoords=None):
        positions = _f32(positions).reshape(-1, 3)
        V = np.zeros((positions.shape[0], 8), np.float32); V[:, 0:3] = positions
        flags = 0
        if normals is not None:
            V[:, 3:6] = _f32(normals).reshape(-1, 3); flags |= 1
        if texcoords is not None:
            V[:, 6:8] = _f32(texcoords).reshape(-1, 2); flags |= 2
        F = np.zeros((np.asarray(faces).reshape(-1, 3).shape[0], 4), np.uint32)
        F[:, :3] = np.asarray(faces, dtype=np.uint32).reshape(-1, 3)
        self.V, self.F, self.flags = V, F, flags
        return self

    def _adopt(self, md):
        try:
            nv, nf = md.vertex_count, md.face_count
            self.V = np.ctypeslib.as_array(md.vertices, shape=(max(nv, 1), 8))[:nv].copy() if nv else np.zeros((0, 8), np.float32)
            self.F = np.ctypeslib.as_array(md.faces, shape=(max(nf, 1), 4))[:nf].copy() if nf else np.zeros((0, 4), np.uint32)
            self.flags = int(md.flags)
        finally:
            lib().har_mesh_free(C.byref(md))
        return self

    @staticmethod
    def _tw(to_world):
        return _fp(to_world.data) if to_world is not None else None

    def from_ply(self, filename, face_normals=False, flip_tex_coords=False, to_world=None, flip_normals=False):
        """PLYMesh (src/shapes/ply.cpp): parsed by the C++ host library into the packed layout; `to_world` / `flip_normals` are
        baked before missing normals are regenerated, as PackedMesh::set_transform does (mesh_utils.cpp:33-44)."""
        md = _capi.HarMeshData()
        check(lib().har_mesh_load_ply(str(filename).encode(), int(bool(face_normals)), int(bool(flip_tex_coords)), self._tw(to_world), int(bool(flip_normals)), C.byref(md)))
        return self._adopt(md)

    def from_obj(self, filename, face_normals=False, flip_tex_coords=True, to_world=None, flip_normals=False):
        """OBJMesh (src/shapes/obj.cpp) + Mesh::from_corners (mesh_utils.cpp:210-560)"""
        md = _capi.HarMeshData()
        check(lib().har_mesh_load_obj(str(filename).encode(), int(bool(face_normals)), int(bool(flip_tex_coords)), self._tw(to_world), int(bool(flip_normals)), C.byref(md)))
        return self._adopt(md)

    def from_serialized(self, filename, shape_index=0, face_normals=None, to_world=None, flip_normals=False):
        """SerializedMesh (src/shapes/serialized.cpp), container versions 3, 4 and 5; face_normals=None: the property is unset (a v5 file's own flag applies)"""
        md = _capi.HarMeshData()
        fn = -1 if face_normals is None else int(bool(face_normals))
        check(lib().har_mesh_load_serialized(str(filename).encode(), int(shape_index), fn, self._tw(to_world), int(bool(flip_normals)), C.byref(md)))
        return self._adopt(md)

    def recompute_vertex_normals(self):
        """Mesh::compute_normals (src/render/mesh.cpp:1218-1267)"""
        self.V = np.ascontiguousarray(self.V); self.F = np.ascontiguousarray(self.F)
        check(lib().har_mesh_compute_normals(self.V.shape[0], _fp(self.V), self.F.shape[0], _up(self.F)))
        self.flags |= 1
        return self

    def transform(self, to_world):
        self.V = np.ascontiguousarray(self.V); self.F = np.ascontiguousarray(self.F)
        check(lib().har_mesh_transform(_fp(to_world.data), self.V.shape[0], _fp(self.V), self.F.shape[0], _up(self.F), self.flags & 1))
        return self

    def face_count(self):
        return self.F.shape[0]

    def vertex_count(self):
        return self.V.shape[0]


# Shape(props) (src/render/shape.cpp:17-60) + Mesh(props) (src/render/mesh.cpp:160-175): rectangle and cube are Mesh plugins in this tree.
# `silhouette_sampling_weight` only feeds the projective integrators' boundary sampling: queried, no effect on `path` / `prb`.
_SHAPE_PROPS = ('to_world', 'flip_normals', 'face_normals', 'silhouette_sampling_weight')
_SHAPE_CHILDREN = ('bsdf', 'emitter')       # Shape(props) (shape.cpp:22-49) also takes sensors, media and texture attributes: not part of this variant, refused


def _rectangle(props):
    _check_props('rectangle', props, _SHAPE_PROPS, unsupported=(('face_normals', False),), children=_SHAPE_CHILDREN)
    tw = props.get('to_world', ScalarTransform4f())
    m = Mesh("rectangle")
    V = np.empty((4, 8), np.float32); F = np.empty((2, 4), np.uint32); n = np.empty(3, np.float32); ia = C.c_float()
    check(lib().har_shape_rectangle(_fp(tw.data), 1 if props.get('flip_normals', False) else 0, _fp(V), _up(F), _fp(n), C.byref(ia)))
    m.V, m.F, m.flags = V, F, 3
    m.rect = dict(to_world=tw, normal=n, inv_area=ia.value, flip=bool(props.get('flip_normals', False)))
    return m


def _cube(props):
    _check_props('cube', props, _SHAPE_PROPS, unsupported=(('face_normals', False), ('flip_normals', False)), children=_SHAPE_CHILDREN)
    tw = props.get('to_world', ScalarTransform4f())
    m = Mesh("cube")
    V = np.empty((24, 8), np.float32); F = np.empty((12, 4), np.uint32)
    check(lib().har_shape_cube(_fp(tw.data), _fp(V), _up(F)))
    m.V, m.F, m.flags = V, F, 3
    return m


# ObjectType of every plugin this variant has (include/mitsuba/core/object.h: ObjectType; PluginManager::create_object checks it, plugin.cpp:258-263).
# `rgb` is the dict form of a colour property, which the reference's loader turns into an `srgb` texture object (src/core/python/parser.cpp) -- a texture here.
_PLUGIN_KINDS = {
    'scene': 'scene', 'path': 'integrator', 'prb': 'integrator', 'perspective': 'sensor', 'orthographic': 'sensor', 'hdrfilm': 'film', 'independent': 'sampler',
    'diffuse': 'bsdf', 'dielectric': 'bsdf', 'conductor': 'bsdf', 'plastic': 'bsdf', 'roughconductor': 'bsdf', 'roughplastic': 'bsdf', 'twosided': 'bsdf',
    'area': 'emitter', 'constant': 'emitter', 'envmap': 'emitter', 'point': 'emitter', 'spot': 'emitter', 'directional': 'emitter',
    'rectangle': 'shape', 'cube': 'shape', 'mesh': 'shape', 'ply': 'shape', 'obj': 'shape', 'serialized': 'shape', 'shapegroup': 'shape', 'instance': 'shape',
    'gaussian': 'rfilter', 'box': 'rfilter', 'tent': 'rfilter', 'mitchell': 'rfilter', 'catmullrom': 'rfilter', 'lanczos': 'rfilter',
    'rgb': 'texture', 'bitmap': 'texture',
}
_ALL_KINDS = tuple(sorted(set(_PLUGIN_KINDS.values())))


def _plugin_kind(t):
    """The ObjectType of plugin `t`; an unknown name is the reference's `Plugin "..." could not be found` (plugin.cpp:189)."""
    k = _PLUGIN_KINDS.get(t)
    if k is None and (t, VARIANT) in _REGISTRY:
        k = 'integrator'                     # mi.register_integrator is the only way a plugin gets into the registry from outside
    if k is None:
        raise RuntimeError("Plugin \"%s\" not found for variant \"%s\" (could not be found). Available: %s" % (t, VARIANT, sorted(_PLUGIN_KINDS)))
    return k


def _object_kind(v):
    """ObjectType of a child value: a dict by its plugin name, an instantiated object by its class; None: not an object (a number, a transform, a `ref`)"""
    if isinstance(v, dict):
        if 'type' not in v or v['type'] == 'ref':
            return None
        return _plugin_kind(v['type'])
    for cls, kind in ((Film, 'film'), (Sampler, 'sampler'), (BSDF, 'bsdf'), (Mesh, 'shape')):
        if isinstance(v, cls):
            return kind
    g = globals()
    for name, kind in (('Sensor', 'sensor'), ('Integrator', 'integrator'), ('ShapeGroup', 'shape'), ('Instance', 'shape'), ('Scene', 'scene'), ('AreaLight', 'emitter'),
                       ('ConstantEmitter', 'emitter'), ('EnvmapEmitter', 'emitter'), ('PointLight', 'emitter'), ('SpotLight', 'emitter'), ('DirectionalEmitter', 'emitter')):
        if name in g and isinstance(v, g[name]):
            return kind
    return None


def _type_mismatch(v, actual, expected):
    name = v.get('type') if isinstance(v, dict) else {'Film': 'hdrfilm', 'Sampler': 'independent'}.get(type(v).__name__, getattr(v, 'kind', None) or getattr(v, 'type', None) or type(v).__name__)
    return RuntimeError("Type mismatch: the instantiated plugin \"%s\" is of type \"%s\", which does not match the expected type \"%s\"." % (name, actual, "|".join(expected)))


def _check_props(plugin, props, known, unsupported=(), free_children=True, children=(), slots=None, slot_kind=('texture',)):
    """The reference's plugin loader rejects properties a plugin never queried ("Unreferenced property", src/core/plugin.cpp / properties.cpp)
    -- a silently ignored property would be a silently different picture.  `unsupported`: (name, neutral value) pairs the reference knows
    but hip_ad_rgb does not implement; anything but the neutral value is refused.

    Child OBJECTS are held to the same rule by their plugin type (round 6): every dict with a `type` names a plugin that must exist
    (plugin.cpp:189) and be of the ObjectType its position wants (plugin.cpp:258-263 "Type mismatch") --
      * under a name of `known`: the kinds of `slots[name]`, else `slot_kind` (object-valued properties of BSDFs / emitters are textures);
      * under a free name (`free_children`: plugins that walk props.objects()): one of `children`; another kind is an object the constructor
        never casts successfully, i.e. an unreferenced property (properties.h:700-725)."""
    for name, neutral in unsupported:
        if name in props and props[name] != neutral:
            raise RuntimeError("%s: property \"%s\" = %r is not implemented by hip_ad_rgb" % (plugin, name, props[name]))
    slots = slots or {}
    for k, v in props.items():
        if k in ('type', 'id'):
            continue
        kind = _object_kind(v)                               # raises for a plugin that does not exist, wherever it sits
        if k in known or any(k == u[0] for u in unsupported):
            want = slots.get(k, slot_kind)
            if kind is not None and kind not in want:
                raise _type_mismatch(v, kind, want)
            continue
        if kind is not None or (isinstance(v, dict) and v.get('type') == 'ref'):
            if not free_children:
                raise RuntimeError("Unreferenced property \"%s\" in plugin of type \"%s\"!" % (k, plugin))
            if kind is not None and kind not in children:
                raise RuntimeError("Unreferenced property \"%s\" in plugin of type \"%s\": an object of type \"%s\" (plugin \"%s\") is not a child this plugin takes (%s)"
                                   % (k, plugin, kind, v.get('type') if isinstance(v, dict) else type(v).__name__, "|".join(children) or "none"))
            continue
        if k.startswith('_arg_'):
            continue
        if free_children and hasattr(v, 'har'):
            continue
        raise RuntimeError("Unreferenced property \"%s\" in plugin of type \"%s\"!" % (k, plugin))


class Sampler:
    """IndependentSampler (src/samplers/independent.cpp) over PCG32Sampler (src/render/sampler.cpp)."""

    def __init__(self, props=None):
        props = props or {}
        if props.get('type', 'independent') != 'independent':        # the other samplers of the reference (stratified, multijitter, orthogonal, ldsampler) draw DIFFERENT streams
            _plugin_kind(props['type'])                               # not a plugin at all: plugin.cpp:189
            raise _type_mismatch(props, _plugin_kind(props['type']), ('sampler',))
        _check_props('independent', props, ('sample_count', 'seed'), free_children=False)
        self.m_sample_count = int(props.get('sample_count', 4))
        self.m_base_seed = int(props.get('seed', 0))
        self.m_samples_per_wavefront = 1
        self.m_wavefront_size = 0
        self.state = None; self.inc = None

    def sample_count(self):
        return self.m_sample_count

    def set_sample_count(self, spp):
        self.m_sample_count = int(spp)

    def set_samples_per_wavefront(self, n):
        self.m_samples_per_wavefront = int(n)
        if self.m_sample_count % self.m_samples_per_wavefront != 0:
            raise RuntimeError("sample_count should be a multiple of samples_per_wavefront!")

    def wavefront_size(self):
        return self.m_wavefront_size

    def seeded(self):
        return self.state is not None

    def clone(self):
        """Sampler::clone (sampler.h:89-99): same configuration and the same streams (har_sampler_clone)"""
        s = Sampler(); s.__dict__.update(self.__dict__)
        if self.state is not None:
            torch = _torch()
            s.state = torch.empty_like(self.state); s.inc = torch.empty_like(self.inc)
            check(lib().har_sampler_clone(self.m_wavefront_size, _ptr(self.state), _ptr(self.inc), _ptr(s.state), _ptr(s.inc), _stream()))
        return s

    def advance(self):
        """Sampler::advance (sampler.h:109-115, independent.cpp:69-72): next sample, dimension index 0; the PCG32 streams are not reseeded"""
        self.m_sample_index = getattr(self, "m_sample_index", 0) + 1
        if self.state is not None:
            check(lib().har_sampler_advance(self.m_wavefront_size, _ptr(self.state), _ptr(self.inc), _stream()))

    def schedule_state(self):
        """Sampler::schedule_state (sampler.h:127): nothing to schedule -- there is no tracing JIT on this path"""

    def fork(self):
        s = Sampler(); s.m_sample_count = self.m_sample_count; s.m_base_seed = self.m_base_seed
        return s

    def seed(self, seed, wavefront_size=None):
        torch = _torch(); dev = _device()
        if wavefront_size is None:
            if self.m_wavefront_size == 0:
                raise RuntimeError("Sampler::seed(): wavefront_size should be specified!")
        else:
            self.m_wavefront_size = int(wavefront_size)
        n = self.m_wavefront_size
        self.state = torch.empty(n, dtype=torch.int64, device=dev); self.inc = torch.empty(n, dtype=torch.int64, device=dev)
        self.m_seed_value = (self.m_base_seed + int(seed)) & 0xffffffff          # the streams' increments are a function of (seed value, lane)
        check(lib().har_sampler_seed(self.m_seed_value, 0, n, _ptr(self.state), _ptr(self.inc), _stream()))

    def _active(self, active):
        if active is None:
            return None
        torch = _torch()
        return torch.as_tensor(active, device=_device()).to(torch.uint8).contiguous()

    def next_1d(self, active=None):
        assert self.seeded()
        torch = _torch(); n = self.m_wavefront_size
        out = torch.empty(n, dtype=torch.float32, device=_device()); a = self._active(active)
        check(lib().har_sampler_next_1d(n, _ptr(self.state), _ptr(self.inc), _ptr(a), _ptr(out), _stream()))
        return out

    def next_2d(self, active=None):
        assert self.seeded()
        torch = _torch(); n = self.m_wavefront_size
        out = torch.empty((2, n), dtype=torch.float32, device=_device()); a = self._active(active)
        check(lib().har_sampler_next_2d(n, _ptr(self.state), _ptr(self.inc), _ptr(a), _ptr(out), _stream()))
        return out


class Film:
    """HDRFilm (src/films/hdrfilm.cpp) + its reconstruction filter."""

    def __init__(self, props=None):
        props = props or {}
        if props.get('type', 'hdrfilm') != 'hdrfilm':                 # specfilm is not part of an RGB variant's path
            raise _type_mismatch(props, _plugin_kind(props['type']), ('film',))
        # hdrfilm.cpp:146-208; file_format / component_format only concern Film::write (har_image_write_*: float32 / half channels)
        _check_props('hdrfilm', props, ('width', 'height', 'crop_offset_x', 'crop_offset_y', 'crop_width', 'crop_height', 'pixel_format', 'file_format', 'component_format',
                                        'sample_border', 'compensate', 'banner'), children=('rfilter',))
        if 'compensate' in props:          # hdrfilm.cpp:218-225: marked as queried, warned about, ignored
            import warnings
            warnings.warn("The \"compensate\" (Kahan-style error-compensated accumulation) parameter has been removed and is now ignored.")
        self.width = int(props.get('width', 768)); self.height = int(props.get('height', 576))
        self.crop_offset_ = (int(props.get('crop_offset_x', 0)), int(props.get('crop_offset_y', 0)))
        self.crop_size_ = (int(props.get('crop_width', self.width)), int(props.get('crop_height', self.height)))
        if min(self.width, self.height, *self.crop_offset_, *self.crop_size_) < 0:
            raise RuntimeError("hdrfilm: sizes and offsets are unsigned")
        # Film::set_crop_window (src/render/film.cpp:90-99)
        if self.crop_offset_[0] + self.crop_size_[0] > self.width or self.crop_offset_[1] + self.crop_size_[1] > self.height:
            raise RuntimeError("Invalid crop window specification: crop_offset(%u, %u) + crop_size(%u, %u) > size(%u, %u)"
                               % (self.crop_offset_ + self.crop_size_ + (self.width, self.height)))
        pf = str(props.get('pixel_format', 'rgb')).lower()          # hdrfilm.cpp:149-176
        formats = {'rgb': (0, False), 'rgba': (0, True), 'luminance': (1, False), 'luminance_alpha': (1, True), 'xyz': (2, False), 'xyza': (2, True)}
        if pf not in formats:
            raise RuntimeError("The \"pixel_format\" parameter must either be equal to \"luminance\", \"luminance_alpha\", \"rgb\", \"rgba\",  \"xyz\", \"xyza\". Found %s." % pf)
        self.pixel_format = pf
        self.colour, self.alpha = formats[pf]        # colour: HAR_PIXEL_RGB / _Y / _XYZ of har_film_develop_format; alpha: FilmFlags::Alpha
        rf = next((v for v in props.values() if isinstance(v, dict) and 'type' in v), {'type': 'gaussian'})     # the film's only child object is its rfilter
        self.rf_param1 = 1.0 / 3.0
        if rf['type'] == 'gaussian':                     # src/rfilters/gaussian.cpp:48-55
            self.rfilter = 1; self.stddev = float(rf.get('stddev', 0.5))
        elif rf['type'] == 'box':
            self.rfilter = 0; self.stddev = 0.5
        elif rf['type'] == 'tent':                       # tent.cpp:48-52
            self.rfilter = 2; self.stddev = float(rf.get('radius', 1.0))
        elif rf['type'] == 'mitchell':                   # mitchell.cpp:50-58
            self.rfilter = 3; self.stddev = float(rf.get('B', 1.0 / 3.0)); self.rf_param1 = float(rf.get('C', 1.0 / 3.0))
        elif rf['type'] == 'catmullrom':
            self.rfilter = 4; self.stddev = 0.0
        elif rf['type'] == 'lanczos':                    # lanczos.cpp:52-55
            self.rfilter = 5; self.stddev = float(int(rf.get('lobes', 3)))
        else:
            raise RuntimeError("Plugin \"%s\" not found for variant hip_ad_rgb (rfilters: box, gaussian, tent, mitchell, catmullrom, lanczos)" % rf['type'])

        # Film::sample_border (film.cpp:29-32): samples are also drawn in a border of rfilter->border_size() = ceil(radius - 1/2 - 2 RayEpsilon) pixels
        # (rfilter.cpp:22) around the crop window; the film keeps its size, the splats of border samples are clipped to it
        radius = {0: 0.5, 1: 4.0 * self.stddev, 2: self.stddev, 3: 2.0, 4: 2.0, 5: self.stddev}[self.rfilter]
        self.sample_border_ = bool(props.get('sample_border', False))
        self.border_size_ = max(0, int(math.ceil(np.float32(np.float32(radius) - np.float32(0.5)) - np.float32(2.0 * 1500.0 * 2.0 ** -24)))) if self.sample_border_ else 0

    def splat_reach(self):
        """film rows a sample can reach above / below its own pixel row: the filter footprint of ImageBlock::put (imageblock.cpp:444-470), ceil(radius - 1/2) -- 0 for
        the box filter (film_footprint, har_path.h)"""
        if self.rfilter == 0:
            return 0
        radius = {1: 4.0 * self.stddev, 2: self.stddev, 3: 2.0, 4: 2.0, 5: self.stddev}[self.rfilter]
        return int(math.ceil(np.float32(np.float32(radius) - np.float32(0.5))))

    def band_rows(self, y0, y1):
        """the rows [lo, hi) of the crop window that samples of the sample-grid rows [y0, y1) splat into (sample border + filter reach)"""
        reach = self.splat_reach()
        return max(0, y0 - self.border_size_ - reach), min(self.crop_size_[1], y1 - self.border_size_ + reach)

    def sample_grid(self):
        """pixels of the lane -> pixel map of render(): crop_size + 2 * border_size with sample_border (integrator.cpp:162-165), else the crop size"""
        return (self.crop_size_[0] + 2 * self.border_size_, self.crop_size_[1] + 2 * self.border_size_)

    def sample_border(self):
        return self.sample_border_

    def size(self):
        return (self.width, self.height)

    def crop_size(self):
        return self.crop_size_

    def crop_offset(self):
        return self.crop_offset_


class Sensor:
    """PerspectiveCamera (src/sensors/perspective.cpp) and OrthographicCamera (src/sensors/orthographic.cpp)."""

    def __init__(self, props):
        self.props = dict(props)
        self.kind = props.get('type', 'perspective')
        # sensor.cpp:24-97, perspective.cpp:137-172, orthographic.cpp:93-97
        if self.kind == 'orthographic':
            _check_props('orthographic', props, ('to_world', 'near_clip', 'far_clip', 'film', 'sampler', 'shutter_open', 'shutter_close'), children=('film', 'sampler'),
                         slots={'film': ('film',), 'sampler': ('sampler',)}, slot_kind=())
        else:
            _check_props('perspective', props, ('to_world', 'fov', 'fov_axis', 'focal_length', 'near_clip', 'far_clip', 'film', 'sampler', 'shutter_open', 'shutter_close', 'focus_distance',
                                                'principal_point_offset_x', 'principal_point_offset_y'), children=('film', 'sampler'),
                         slots={'film': ('film',), 'sampler': ('sampler',)}, slot_kind=())
        # child objects are recognised by their class, whatever the property is called (XML children are anonymous: `_arg_0`, ...)
        film = next((v for v in props.values() if isinstance(v, Film)), props.get('film'))
        sampler = next((v for v in props.values() if isinstance(v, Sampler)), props.get('sampler'))
        self.m_film = film if isinstance(film, Film) else Film(film)
        self.m_sampler = sampler if isinstance(sampler, Sampler) else Sampler(sampler)
        self.to_world = props.get('to_world', ScalarTransform4f())
        if self.kind != 'orthographic' and self.to_world.has_scale():            # perspective.cpp:143-146; an orthographic camera's scale sets the size of its view
            raise RuntimeError("Scale factors in the camera-to-world transformation are not allowed!")
        if 'fov' in props and 'focal_length' in props:
            raise RuntimeError("Please specify either a focal length ('focal_length') or a field of view ('fov')!")
        self.near_clip = float(props.get('near_clip', 1e-2)); self.far_clip = float(props.get('far_clip', 1e4))
        self.update()

    def update(self):
        f = self.m_film
        if self.kind == 'orthographic':            # OrthographicCamera::update_camera_transforms (orthographic.cpp:104-121)
            s = _capi.HarSensor()
            if lib().har_orthographic_sensor(_fp(self.to_world.data), self.near_clip, self.far_clip, f.width, f.height, f.crop_offset_[0], f.crop_offset_[1],
                                             f.crop_size_[0], f.crop_size_[1], f.rfilter, f.stddev, C.byref(s)):
                raise RuntimeError("invalid sensor parameters")
            s.rfilter_param1 = f.rf_param1
            s.sample_border = 1 if f.sample_border_ else 0
            self.har = s
            return
        fov_axis = self.props.get('fov_axis', 'x')
        if 'fov' in self.props:
            fov = float(self.props['fov'])
        else:
            fl = str(self.props.get('focal_length', '50mm'))
            value = float(fl[:-2] if fl.endswith('mm') else fl)
            fov = 2.0 * math.degrees(math.atan(math.sqrt(36 * 36 + 24 * 24) / (2.0 * value))); fov_axis = 'diagonal'
        s = _capi.HarSensor()
        rc = lib().har_perspective_sensor(_fp(self.to_world.data), fov, fov_axis.encode(), self.near_clip, self.far_clip,
                                          f.width, f.height, f.crop_offset_[0], f.crop_offset_[1], f.crop_size_[0], f.crop_size_[1],
                                          f.rfilter, f.stddev, C.byref(s))
        s.rfilter_param1 = f.rf_param1
        s.sample_border = 1 if f.sample_border_ else 0
        # perspective.cpp:147-150: the principal point, as a fraction of the film size
        s.principal_point_offset_x = float(self.props.get('principal_point_offset_x', 0.0)); s.principal_point_offset_y = float(self.props.get('principal_point_offset_y', 0.0))
        if rc == 2:
            raise RuntimeError("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
        if rc == 3:
            raise RuntimeError("The horizontal field of view must be in the range [0, 180]!")
        if rc:
            raise RuntimeError("invalid sensor parameters")
        self.har = s

    def x_fov(self):
        """the horizontal field of view in degrees: parse_fov (src/render/sensor.cpp:142-195) of the sensor's `fov` / `fov_axis` / `focal_length` at the film's aspect ratio --
        what PerspectiveCamera::traverse registers as 'x_fov' (perspective.cpp:157)"""
        w, h = self.m_film.width, self.m_film.height
        aspect = w / float(h)
        axis = str(self.props.get('fov_axis', 'x')).lower()
        if 'fov' in self.props:
            fov = float(self.props['fov'])
            if axis == 'smaller':
                axis = 'y' if aspect > 1 else 'x'
            elif axis == 'larger':
                axis = 'x' if aspect > 1 else 'y'
        else:
            fl = str(self.props.get('focal_length', '50mm'))
            fov = 2.0 * math.degrees(math.atan(math.sqrt(36 * 36 + 24 * 24) / (2.0 * float(fl[:-2] if fl.endswith('mm') else fl)))); axis = 'diagonal'
        if axis == 'x':
            return fov
        if axis == 'y':
            return math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
        width = 2.0 * math.tan(0.5 * math.radians(fov)) / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        return math.degrees(2.0 * math.atan(width * 0.5))

    def film(self):
        return self.m_film

    def sampler(self):
        return self.m_sampler

    def sample_ray(self, time, sample1, sample2, sample3=None):
        """PerspectiveCamera::sample_ray: sample2 is the film position in [0,1]^2, tensor [2, n]."""
        torch = _torch(); dev = _device()
        p = torch.as_tensor(sample2, dtype=torch.float32, device=dev).reshape(2, -1).contiguous(); n = p.shape[1]
        o = torch.empty((3, n), dtype=torch.float32, device=dev); d = torch.empty_like(o); mt = torch.empty(n, dtype=torch.float32, device=dev)
        check(lib().har_sensor_sample_ray(C.byref(self.har), n, _ptr(p[0]), _ptr(p[1]), _ptr(o), _ptr(d), _ptr(mt), _stream()))
        return Ray3f(o, d, mt), torch.ones(n, device=dev)


# include/mitsuba/render/ior.h:24-48
_IOR = {"vacuum": 1.0, "helium": 1.000036, "hydrogen": 1.000132, "air": 1.000277, "carbon dioxide": 1.00045, "water": 1.3330,
        "acetone": 1.36, "ethanol": 1.361, "carbon tetrachloride": 1.461, "glycerol": 1.4729, "benzene": 1.501, "silicone oil": 1.52045,
        "bromine": 1.661, "water ice": 1.31, "fused quartz": 1.458, "pyrex": 1.470, "acrylic glass": 1.49, "polypropylene": 1.49,
        "bk7": 1.5046, "sodium chloride": 1.544, "amber": 1.55, "pet": 1.5750, "diamond": 2.419}


def _lookup_ior(props, name, default):
    v = props.get(name, default)
    if isinstance(v, str):
        if v.lower() not in _IOR:
            raise RuntimeError("Could not find a material named \"%s\"" % v)
        return float(_IOR[v.lower()])
    return float(v)


def _rgb_value(v, default, bounded=True, uniform=None):
    """Properties colour value: float, list or {'type': 'rgb', 'value': ...} -> float32[3].  `uniform` = the quantity's name where the reference itself refuses a
    spatially varying texture (point.cpp:80-81, spot.cpp:96-97, directional.cpp:84-85, constant.cpp:60-61): same words"""
    if v is None:
        v = default
    if isinstance(v, dict):
        if v.get('type') != 'rgb' and uniform:
            raise RuntimeError("Expected a non-spatially varying %s spectra!" % uniform)
        if v.get('type') != 'rgb':
            raise RuntimeError("only `rgb` values are implemented for this parameter")
        v = v['value']
    out = _f32([v] * 3 if np.isscalar(v) else list(v))
    if bounded and (np.any(out < 0) or np.any(out > 1)):
        raise RuntimeError("Invalid RGB reflectance value %s, must be in the range [0, 1]!" % out)
    return out


# the properties each BSDF plugin queries (src/bsdfs/*.cpp constructors; MicrofacetDistribution(props), microfacet.h:103-144)
_BSDF_PROPS = {'diffuse': ('reflectance',),
               'dielectric': ('int_ior', 'ext_ior', 'specular_reflectance', 'specular_transmittance'),
               'conductor': ('material', 'eta', 'k', 'specular_reflectance'),
               'plastic': ('int_ior', 'ext_ior', 'diffuse_reflectance', 'specular_reflectance', 'nonlinear'),
               'roughconductor': ('material', 'eta', 'k', 'distribution', 'sample_visible', 'alpha', 'alpha_u', 'alpha_v', 'specular_reflectance'),
               'roughplastic': ('distribution', 'sample_visible', 'alpha', 'alpha_u', 'alpha_v', 'int_ior', 'ext_ior', 'diffuse_reflectance', 'specular_reflectance', 'nonlinear')}
# BitmapTexture(props) (src/textures/bitmap.cpp:175-260); `raw` only silences a range warning for float data in RGB variants, `accel` picks Dr.Jit's texture
# hardware path on CUDA (no effect on values), `format` = the storage type of the texels ("auto" / "variant" / "fp16": this path keeps float32)
_BITMAP_PROPS = ('filename', 'bitmap', 'data', 'to_uv', 'raw', 'accel', 'filter_type', 'wrap_mode', 'format')
BSDF_TYPES = {'diffuse': 0, 'dielectric': 1, 'roughconductor': 2, 'roughplastic': 3, 'conductor': 4, 'plastic': 5}
# (parameter name of slot 0, default), (parameter name of slot 1, default)
_BSDF_SLOTS = {'diffuse': (('reflectance', 0.5), None), 'dielectric': (('specular_reflectance', 1.0), ('specular_transmittance', 1.0)),
               'roughconductor': (('specular_reflectance', 1.0), None), 'roughplastic': (('diffuse_reflectance', 0.5), ('specular_reflectance', 1.0)),
               'conductor': (('specular_reflectance', 1.0), None), 'plastic': (('diffuse_reflectance', 0.5), ('specular_reflectance', 1.0))}


def _bitmap_from_props(refl):
    """BitmapTexture(props) -> (H x W x 3 float32 texels, HarTexture::mode, to_uv rows or None)"""
    # BitmapTexture (src/textures/bitmap.cpp:175-260): texels from a tensor (`data`), a Bitmap object (`bitmap`) or a file (`filename`:
    # OpenEXR / PFM, read by the C++ host library).  Float data is linear; in RGB variants `raw` only silences the [0, 1] range warning.
    if sum(k in refl for k in ('data', 'bitmap', 'filename')) != 1:
        raise RuntimeError("bitmap: exactly one of 'filename', 'bitmap' and 'data' must be specified")
    # filter_type / wrap_mode (bitmap.cpp:182-206) -> HarTexture::mode
    ft = str(refl.get('filter_type', 'bilinear')); wm = str(refl.get('wrap_mode', 'repeat'))
    if ft not in ('nearest', 'bilinear'):
        raise RuntimeError("Invalid filter type \"%s\", must be one of: \"nearest\", or \"bilinear\"!" % ft)
    if wm not in ('repeat', 'mirror', 'clamp'):
        raise RuntimeError("Invalid wrap mode \"%s\", must be one of: \"repeat\", \"mirror\", or \"clamp\"!" % wm)
    tex_mode = (1 if ft == 'nearest' else 0) | {'repeat': 0, 'mirror': 2, 'clamp': 4}[wm]
    _check_props('bitmap', refl, _BITMAP_PROPS, unsupported=(('format', 'auto'),), free_children=False)
    # to_uv (bitmap.cpp:175): uv = m_transform * si.uv before every lookup (:565,792,831,847) -> HarTexture::to_uv, row-major 2 x 3
    tex_to_uv = None
    if 'to_uv' in refl:
        tuv = refl['to_uv']
        if isinstance(tuv, ScalarTransform4f):        # Properties::get<AffineTransform3f> of a stored 4 x 4 (an XML <transform>): Transform::extract (transform.h:441-456)
            m4 = tuv.matrix; tuv = ScalarTransform3f([[m4[0, 0], m4[0, 1], m4[0, 3]], [m4[1, 0], m4[1, 1], m4[1, 3]], [0.0, 0.0, 1.0]])
        if not isinstance(tuv, ScalarTransform3f):
            raise RuntimeError("bitmap: 'to_uv' must be a ScalarTransform3f")
        if abs(float(np.linalg.det(tuv.matrix[:2, :2].astype(np.float64)))) == 0.0:
            raise RuntimeError("bitmap: 'to_uv' is singular")
        tex_to_uv = tuv.rows_2x3()
    if 'data' in refl:
        t = refl['data']
        if hasattr(t, 'detach'):
            t = t.detach().cpu().numpy()
    else:
        b = refl['bitmap'] if 'bitmap' in refl else Bitmap(refl['filename'])
        if not isinstance(b, Bitmap):
            raise RuntimeError("Property \"bitmap\" must be a Bitmap instance.")
        t = b.data
    t = _f32(t)
    if t.ndim == 2:
        t = t[:, :, None]
    if t.ndim != 3 or t.shape[2] not in (1, 3, 4):
        raise RuntimeError("Bitmap raw tensor has dimension %d, expected 3 (H x W x {1, 3, 4})" % t.ndim)
    t = np.repeat(t, 3, axis=2) if t.shape[2] == 1 else t[:, :, :3]
    return np.ascontiguousarray(t, np.float32), tex_mode, tex_to_uv


class BSDF:
    """diffuse / dielectric / conductor / plastic / roughconductor / roughplastic (src/bsdfs/*.cpp), optionally wrapped by `twosided`.
    Colour parameters live in two slots (include/hip_ad_rgb.h HarBSDF); slot 0 may be a raw `bitmap`."""

    def __init__(self, props=None, id=None):
        props = props or {}
        self.id = id
        self.kind = props.get('type', 'diffuse')
        if self.kind not in BSDF_TYPES:
            raise RuntimeError("Plugin \"%s\" not found for variant hip_ad_rgb" % self.kind)
        _check_props(self.kind, props, _BSDF_PROPS[self.kind], free_children=False)
        (name0, def0), slot1 = _BSDF_SLOTS[self.kind]
        self.slot0_name = name0; self.slot1_name = slot1[0] if slot1 else None
        refl = props.get(name0, {'type': 'rgb', 'value': [def0] * 3})
        if isinstance(refl, (int, float)):
            refl = {'type': 'rgb', 'value': [refl] * 3}
        self.texture = None; self.tex_mode = 0; self.tex_to_uv = None
        if refl['type'] == 'rgb':
            self.value = _rgb_value(refl, def0)
        elif refl['type'] == 'bitmap':
            self.texture, self.tex_mode, self.tex_to_uv = _bitmap_from_props(refl)
            self.value = _f32([0.5, 0.5, 0.5])
        else:
            raise RuntimeError("Plugin \"%s\" not found for variant hip_ad_rgb (textures: rgb, bitmap)" % refl['type'])
        self.value2 = _rgb_value(props.get(slot1[0]), slot1[1]) if slot1 else _f32([0, 0, 0])
        self.flags = 0; self.alpha_u = self.alpha_v = 0.1; self.eta = 1.0; self.anisotropic = False
        self.eta_c = _f32([0, 0, 0]); self.k_c = _f32([1, 1, 1]); self.back = None
        if self.kind in ('roughconductor', 'roughplastic'):          # MicrofacetDistribution(props), microfacet.h:103-144
            distr = str(props.get('distribution', 'beckmann')).lower()
            if distr not in ('beckmann', 'ggx'):
                raise RuntimeError("Specified an invalid distribution \"%s\", must be \"beckmann\" or \"ggx\"!" % distr)
            if distr == 'ggx':
                self.flags |= 2
            if props.get('sample_visible', True):
                self.flags |= 4
            if 'alpha_u' in props or 'alpha_v' in props:
                if not ('alpha_u' in props and 'alpha_v' in props):
                    raise RuntimeError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
                if 'alpha' in props:
                    raise RuntimeError("Microfacet model: please specifyeither 'alpha' or 'alpha_u'/'alpha_v'.")
                self.alpha_u = float(props['alpha_u']); self.alpha_v = float(props['alpha_v']); self.anisotropic = True
            else:
                self.alpha_u = self.alpha_v = float(props.get('alpha', 0.1))
        if self.kind in ('roughconductor', 'conductor'):             # roughconductor.cpp:163-172, conductor.cpp:228-237
            if props.get('material', 'none') != 'none':
                if 'eta' in props:
                    raise RuntimeError("Should specify either (eta, k) or material, not both.")
                raise RuntimeError("%s: `material` presets need the spectral IOR data files, give (eta, k) instead" % self.kind)
            self.eta_c = _rgb_value(props.get('eta'), 0.0, bounded=False); self.k_c = _rgb_value(props.get('k'), 1.0, bounded=False)
        if self.kind in ('dielectric', 'roughplastic', 'plastic'):
            int_ior = _lookup_ior(props, 'int_ior', 'bk7' if self.kind == 'dielectric' else 'polypropylene')
            ext_ior = _lookup_ior(props, 'ext_ior', 'air')
            if int_ior < 0 or ext_ior < 0 or (self.kind == 'roughplastic' and int_ior == ext_ior):
                raise RuntimeError("The interior and exterior indices of refraction must be positive" + (" and differ!" if self.kind == 'roughplastic' else "!"))
            self.eta = float(np.float32(int_ior) / np.float32(ext_ior))
            if self.kind == 'roughplastic' and self.alpha_u != self.alpha_v:
                raise RuntimeError("The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!")
            if self.kind in ('roughplastic', 'plastic') and props.get('nonlinear', False):
                self.flags |= 8
        self.scene = None; self.index = None

    def _bind(self):
        if self.scene is None:          # stand-alone BSDF: private scene holding just this plugin
            Scene({'_bsdf': self})
        return self.scene._handle(), self.index

    def _prep(self, si_wi, si_uv, n):
        torch = _torch(); dev = _device()
        wi = torch.as_tensor(si_wi, dtype=torch.float32, device=dev).reshape(3, -1)
        wi = wi.expand(3, n).contiguous() if wi.shape[1] != n else wi.contiguous()
        uv = torch.zeros((2, n), dtype=torch.float32, device=dev) if si_uv is None else torch.as_tensor(si_uv, dtype=torch.float32, device=dev).reshape(2, -1).contiguous()
        return wi, uv

    def _eval(self, which, ctx, si, wo, active):
        torch = _torch(); dev = _device(); h, idx = self._bind()
        wo = torch.as_tensor(wo, dtype=torch.float32, device=dev).reshape(3, -1).contiguous(); n = wo.shape[1]
        wi, uv = self._prep(si.wi, getattr(si, 'uv', None), n)
        a = _mask(active, n); c = _ctx(ctx)
        val = torch.empty((3, n), dtype=torch.float32, device=dev) if which != 'pdf' else None
        pdf = torch.empty(n, dtype=torch.float32, device=dev) if which != 'eval' else None
        if which == 'eval_pdf':
            check(lib().har_bsdf_eval_pdf(h, idx, c, n, _ptr(wi), _ptr(uv), _ptr(wo), _ptr(a), _ptr(val), _ptr(pdf), _stream()))
        elif which == 'eval':
            check(lib().har_bsdf_eval(h, idx, c, n, _ptr(wi), _ptr(uv), _ptr(wo), _ptr(a), _ptr(val), _stream()))
        else:
            check(lib().har_bsdf_pdf(h, idx, c, n, _ptr(wi), _ptr(uv), _ptr(wo), _ptr(a), _ptr(pdf), _stream()))
        return val, pdf

    def eval_pdf(self, ctx, si, wo, active=True):
        """BSDF::eval_pdf(ctx, si, wo, active) (bsdf.h:431-465)"""
        return self._eval('eval_pdf', ctx, si, wo, active)

    def eval(self, ctx, si, wo, active=True):
        return self._eval('eval', ctx, si, wo, active)[0]

    def pdf(self, ctx, si, wo, active=True):
        return self._eval('pdf', ctx, si, wo, active)[1]

    def sample(self, ctx, si, sample1, sample2, active=True):
        """BSDF::sample(ctx, si, sample1, sample2, active) -> (BSDFSample3f, weight) (bsdf.h:322-373)"""
        torch = _torch(); dev = _device(); h, idx = self._bind()
        s2 = torch.as_tensor(sample2, dtype=torch.float32, device=dev).reshape(2, -1).contiguous(); n = s2.shape[1]
        wi, uv = self._prep(si.wi, getattr(si, 'uv', None), n)
        wo = torch.empty((3, n), dtype=torch.float32, device=dev); pdf = torch.empty(n, dtype=torch.float32, device=dev); w = torch.empty_like(wo)
        s1 = torch.as_tensor(sample1, dtype=torch.float32, device=dev).reshape(-1)
        s1 = (s1.expand(n) if s1.numel() != n else s1).contiguous()
        eta = torch.empty(n, dtype=torch.float32, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev); sc = torch.empty(n, dtype=torch.int32, device=dev)
        a = _mask(active, n)
        check(lib().har_bsdf_sample(h, idx, _ctx(ctx), n, _ptr(wi), _ptr(uv), _ptr(s1), _ptr(s2), _ptr(a), _ptr(wo), _ptr(pdf), _ptr(w), _ptr(eta), _ptr(st), _ptr(sc), _stream()))
        bs = type("BSDFSample3f", (), dict(wo=wo, pdf=pdf, eta=eta, sampled_type=st, sampled_component=sc, delta=(st & BSDFFlags.Delta) != 0))()
        return bs, w

    def component_count(self):
        """BSDF::component_count(): the lobes of this plugin (twosided: the front BSDF's, then the back's, twosided.cpp:86-99)"""
        own = {'dielectric': 2, 'roughplastic': 2, 'plastic': 2}
        n = own.get(self.kind, 1)
        if self.flags & 1:
            return n + (own.get(self.back.kind, 1) if self.back is not None else n)
        return n


def _mk_twosided(props, named, key):
    """TwoSidedBRDF (src/bsdfs/twosided.cpp:70-110): one or two nested BSDFs without a transmission component."""
    _check_props('twosided', props, ('allow_transmission',), unsupported=(('allow_transmission', False),), children=('bsdf',))      # twosided.cpp:76-104: nested BSDFs under any name
    nested = []
    for k, v in props.items():
        if k in ('type', 'id', 'allow_transmission'):
            continue
        obj = _resolve(v, named, k) if isinstance(v, dict) else v
        if isinstance(obj, BSDF):
            nested.append(obj)
    if not nested or len(nested) > 2:
        raise RuntimeError("twosided: a maximum of two nested BSDFs can be specified (and at least one)!")
    for b in nested:
        if b.kind == 'dielectric':
            raise RuntimeError("Only materials without a transmission component can be nested!")
    front = nested[0]
    if front.flags & 1:
        raise RuntimeError("twosided: nested BSDF is already two-sided")
    front.flags |= 1; front.id = front.id or key
    front.back = nested[1] if len(nested) == 2 and nested[1] is not nested[0] else None
    if front.back is not None:
        front.back.twosided_parent = front          # (its parameters are '<twosided>.brdf_1.*': Scene._bsdf_key_base)
    if key:
        front.id = key
    return front


def _sampling_weight(props):
    """Emitter(props) (src/render/emitter.cpp:9): `sampling_weight`, default 1 -- Scene::update_emitter_sampling_distribution (scene.cpp:120-141) switches from the
    uniform emitter choice to a DiscreteDistribution over the weights as soon as one of them is not 1"""
    w = float(np.float32(props.get('sampling_weight', 1.0)))
    if not (w >= 0.0) or not math.isfinite(w):
        raise RuntimeError("DiscreteDistribution: entries must be non-negative!")
    return w


def _emissive_rgb(plugin, name, v):
    """Properties::get_emissive_texture in RGB variants: a float, an `rgb` colour (unbounded) -- spatially varying textures are refused where the path needs the
    texture-importance-sampling branch (area.cpp:133-165) or is not written for them"""
    if isinstance(v, dict) and v.get('type') not in (None, 'rgb') and plugin == 'constant':
        raise RuntimeError("Expected a non-spatially varying radiance spectra!")           # the reference's own refusal: constant.cpp:60-61
    if isinstance(v, dict) and v.get('type') not in (None, 'rgb'):
        raise RuntimeError("%s: a spatially varying \"%s\" (texture plugin \"%s\") is not implemented by hip_ad_rgb -- the reference then importance-samples the "
                           "texture and maps the sample through Shape::eval_parameterization (src/emitters/area.cpp:133-165); use an `rgb` value" % (plugin, name, v.get('type')))
    return _rgb_value(v, 1.0, bounded=False)


def _check_sampling_transform(rows):
    """BitmapTextureImpl::check_sampling_transform (src/textures/bitmap.cpp:976-992): position sampling needs a to_uv that maps the unit square's corners onto themselves"""
    m = np.asarray(rows if rows is not None else [1, 0, 0, 0, 1, 0], np.float32).reshape(2, 3)
    corners = np.asarray([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    hits = 0
    for c in corners:
        q = m[:, :2] @ c + m[:, 2]
        for j, d in enumerate(corners):
            if float(((q - d) ** 2).sum()) < 1e-8:
                hits |= 1 << j
    if hits != 0xF:
        raise RuntimeError("Bitmap texture: position sampling (e.g. of an area emitter's radiance) requires a 'to_uv' transformation that maps the unit square onto "
                           "itself, such as a flip, a transpose or a multiple of a 90 degree rotation.")


class AreaLight:
    """AreaLight (src/emitters/area.cpp): a uniform `radiance`, or a `bitmap` on a rectangle; it inherits the placement of its parent shape (a `to_world` is an error, :66-69)."""

    def __init__(self, props):
        if 'to_world' in props:
            raise RuntimeError("Found a 'to_world' transformation -- this is not allowed. The area light inherits this transformation from its parent shape.")
        _check_props('area', props, ('radiance', 'sampling_weight'), free_children=False)
        rad = props.get('radiance', {'type': 'rgb', 'value': 1.0})
        self.texture = None; self.tex_mode = 0; self.tex_to_uv = None; self.tex_index = None
        if isinstance(rad, dict) and rad.get('type') == 'bitmap':
            # a spatially varying radiance (is_spatially_varying(), area.cpp:74-75): the texture is importance-sampled and mapped onto the shape by
            # Shape::eval_parameterization (:133-165) -- built for rectangles (HarEmitter type 7); the parent shape decides (Scene._add_mesh)
            self.texture, self.tex_mode, self.tex_to_uv = _bitmap_from_props(rad)
            _check_sampling_transform(self.tex_to_uv)
            self.radiance = _f32([0.0, 0.0, 0.0])
        else:
            self.radiance = _emissive_rgb('area', 'radiance', rad)       # get_emissive_texture("radiance", 1.f), :71
        self.sampling_weight = _sampling_weight(props)


class ConstantEmitter:
    """ConstantBackgroundEmitter (src/emitters/constant.cpp): uniform environment radiance."""

    def __init__(self, props):
        _check_props('constant', props, ('radiance', 'sampling_weight', 'to_world'), free_children=False)      # Endpoint(props) queries to_world (endpoint.cpp:12); a uniform environment does not depend on it
        self.radiance = _emissive_rgb('constant', 'radiance', props.get('radiance', {'type': 'rgb', 'value': 1.0}))
        self.sampling_weight = _sampling_weight(props)


class PointLight:
    """PointLight (src/emitters/point.cpp): isotropic point source of radiant `intensity` at `position` (or the translation of `to_world`);
    EmitterFlags::DeltaPosition -- only emitter sampling finds it, with MIS weight 1 (path.cpp:274, prb.py:211)."""

    def __init__(self, props):
        _check_props("point", props, ('position', 'to_world', 'intensity', 'sampling_weight'), free_children=False)
        self.sampling_weight = _sampling_weight(props)
        if 'position' in props:
            if 'to_world' in props:                                  # point.cpp:65-68
                raise RuntimeError("Only one of the parameters 'position' and 'to_world' can be specified at the same time!'")
            self.position = _f32(list(props['position'])).reshape(3)
        else:
            self.position = _f32(props.get('to_world', ScalarTransform4f()).col_major_3x4()[9:12])      # m_to_world.translation(), point.cpp:72
        self.intensity = _rgb_value(props.get('intensity', {'type': 'rgb', 'value': 1.0}), 1.0, bounded=False, uniform='intensity')   # get_emissive_texture("intensity", 1.f), :77


class SpotLight:
    """SpotLight (src/emitters/spot.cpp) without a `texture`: a point source at the origin of `to_world` that radiates `intensity` along the local +z axis inside
    `beam_width` degrees, falling off linearly in the angle to zero at `cutoff_angle` (falloff_curve, :143-151); EmitterFlags::DeltaPosition."""

    def __init__(self, props):
        _check_props("spot", props, ('to_world', 'intensity', 'cutoff_angle', 'beam_width', 'texture', 'sampling_weight'), free_children=False)
        self.sampling_weight = _sampling_weight(props)
        if 'texture' in props:
            raise RuntimeError("spot: property \"texture\" is not implemented by hip_ad_rgb")
        self.to_world = props.get('to_world', ScalarTransform4f())
        self.intensity = _rgb_value(props.get('intensity', {'type': 'rgb', 'value': 1.0}), 1.0, bounded=False, uniform='intensity')      # get_emissive_texture("intensity", 1.f), :93
        self.cutoff_angle = float(np.float32(props.get('cutoff_angle', 20.0)))                                        # :105-107
        self.beam_width = float(np.float32(props.get('beam_width', np.float32(self.cutoff_angle) * np.float32(3.0) / np.float32(4.0))))
        if not (self.cutoff_angle >= self.beam_width and self.cutoff_angle > 0):
            raise RuntimeError("spot: cutoff_angle must be positive and not smaller than beam_width")


class DirectionalEmitter:
    """DirectionalEmitter (src/emitters/directional.cpp): a distant source that delivers `irradiance` onto surfaces perpendicular to its direction of travel,
    the +z axis of `to_world` (or `direction`, lowered to look_at(0, direction, up) as in :69-78); EmitterFlags::Infinite | DeltaDirection -- sampled with
    MIS weight 1, invisible to escaping rays."""

    def __init__(self, props):
        _check_props("directional", props, ('to_world', 'direction', 'irradiance', 'sampling_weight'), free_children=False)
        self.sampling_weight = _sampling_weight(props)
        if 'direction' in props:
            if 'to_world' in props:                                   # directional.cpp:70-72
                raise RuntimeError("Only one of the parameters 'direction' and 'to_world' can be specified at the same time!'")
            d = _f32(list(props['direction'])).reshape(3); d = (d / np.sqrt(np.float32((d * d).sum()))).astype(np.float32)
            # coordinate_system(direction).first (vector.h:118-138, Duff et al.) in float32
            sign = np.float32(np.copysign(1.0, d[2])); a = np.float32(-1.0) / (sign + d[2]); b = d[0] * d[1] * a
            up = _f32([np.copysign(1.0, d[2]) * (d[0] * d[0] * a) + 1.0, np.copysign(1.0, d[2]) * b, -np.copysign(1.0, d[2]) * d[0]])
            self.to_world = ScalarTransform4f().look_at(origin=[0.0, 0.0, 0.0], target=[float(x) for x in d], up=[float(x) for x in up])
        else:
            self.to_world = props.get('to_world', ScalarTransform4f())
        self.irradiance = _rgb_value(props.get('irradiance', {'type': 'rgb', 'value': 1.0}), 1.0, bounded=False, uniform='irradiance')      # get_emissive_texture("irradiance", 1.f), :80


class EnvmapEmitter:
    """EnvironmentMapEmitter (src/emitters/envmap.cpp): lat-long radiance image, importance sampled by luminance * sin(theta)."""

    def __init__(self, props):
        _check_props('envmap', props, ('filename', 'bitmap', 'scale', 'mis_compensation', 'to_world', 'sampling_weight'), free_children=False)      # envmap.cpp:118-200
        self.sampling_weight = _sampling_weight(props)
        if 'bitmap' in props:
            if 'filename' in props:
                raise RuntimeError("Cannot specify both \"bitmap\" and \"filename\".")
            b = props['bitmap']
            if not isinstance(b, Bitmap):
                raise RuntimeError("Property \"bitmap\" must be a Bitmap instance.")
        elif 'filename' in props:
            b = Bitmap(props['filename'])
        else:
            raise RuntimeError("envmap: one of \"filename\" / \"bitmap\" is required")
        a = b.data
        if a.shape[2] == 1:                                  # Bitmap::convert(PixelFormat::RGB): luminance is replicated
            a = np.repeat(a, 3, axis=2)
        elif a.shape[2] == 4:
            a = a[:, :, :3]
        elif a.shape[2] != 3:
            raise RuntimeError("envmap: unsupported channel count %d" % a.shape[2])
        self.data = np.ascontiguousarray(a, np.float32)
        self.scale = float(props.get('scale', 1.0))
        self.mis_compensation = bool(props.get('mis_compensation', False))
        self.to_world = props.get('to_world', ScalarTransform4f())


class ShapeGroup:
    def __init__(self, shapes, keys=None):
        self.shapes = shapes
        self.keys = list(keys) if keys else [str(i) for i in range(len(shapes))]      # the children's names: '<group>.<child>' in mi.traverse, as in the reference


class Instance:
    def __init__(self, group, to_world):
        self.group = group; self.to_world = to_world


class Integrator:
    """PathIntegrator (src/integrators/path.cpp) / PRBIntegrator (ad/integrators/prb.py)."""

    def __init__(self, props):
        self.type = props['type']
        # integrator.cpp:26-33,128-147,539-550; block_size only shapes the scalar / LLVM-parallel drivers; the last four are hip_ad_rgb extensions
        _check_props(self.type, props, ('max_depth', 'rr_depth', 'hide_emitters', 'samples_per_pass', 'block_size', 'chunk_lanes', 'replay_cache', 'material_queues', 'packet_tracing', 'emitter_gradients', 'shape_gradients', 'bsdf_parameter_gradients', 'light_texel_gradients'),
                     unsupported=(('timeout', -1.0),), free_children=False, slot_kind=())
        default_depth = -1 if self.type == 'path' else 6        # integrator.cpp:539, common.py:31
        self.max_depth = int(props.get('max_depth', default_depth))
        self.rr_depth = int(props.get('rr_depth', 5))
        self.hide_emitters = bool(props.get('hide_emitters', False))      # integrator.cpp:29
        if self.max_depth < 0 and self.max_depth != -1:
            raise RuntimeError("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        if self.rr_depth <= 0:
            raise RuntimeError("\"rr_depth\" must be set to a value greater than zero!")
        self.chunk_lanes = int(props.get('chunk_lanes', 0))
        self.replay_cache = bool(props.get('replay_cache', True))      # hip_ad_rgb extension, see har_integrator_set_replay_cache
        self.material_queues = bool(props.get('material_queues', False))   # hip_ad_rgb extension, see har_integrator_set_material_queues
        # hip_ad_rgb extension, see har_integrator_set_packet_tracing: None = automatic, False / True force the wave-shared descent of the camera rays off / on
        self.packet_tracing = props.get('packet_tracing', None)
        self.emitter_gradients = bool(props.get('emitter_gradients', True))   # d / d radiance of area / constant emitters (har_integrator_set_grad_emitters)
        # d / d vertex positions (har_integrator_set_grad_positions): False, True (every eligible mesh) or a list of '<shape>.positions' keys --
        # the stand-in for dr.enable_grad(params[key]) of the reference
        self.shape_gradients = props.get('shape_gradients', False)
        # gradients of `alpha` / `alpha_u` / `alpha_v`, `eta`, `k` (roughconductor) and `alpha`, `specular_reflectance` (roughplastic): har_integrator_set_grad_bsdf_params
        self.bsdf_parameter_gradients = bool(props.get('bsdf_parameter_gradients', False))
        # gradients w.r.t. the texels of a bitmap `radiance` of area lights ('<shape>.emitter.radiance.data'; area.cpp:64-70): har_integrator_set_grad_light_texels
        self.light_texel_gradients = bool(props.get('light_texel_gradients', False))
        # SamplingIntegrator property (integrator.cpp:140-147); the Python AD integrators do not query it, and an
        # unqueried property is an error in the reference's plugin loader
        self.samples_per_pass = props.get('samples_per_pass', None)
        if self.samples_per_pass is not None:
            if self.type != 'path':
                raise RuntimeError("Unreferenced property \"samples_per_pass\" in plugin of type \"%s\"" % self.type)
            self.samples_per_pass = int(self.samples_per_pass)
        self._h = None

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            check(lib().har_integrator_create(0 if self.type == 'path' else 1, self.max_depth, self.rr_depth, self.chunk_lanes, C.byref(h)))
            self._h = h
            if not self.replay_cache:
                check(lib().har_integrator_set_replay_cache(h, 0))
            if self.material_queues:
                check(lib().har_integrator_set_material_queues(h, 1))
            if self.packet_tracing is not None:
                check(lib().har_integrator_set_packet_tracing(h, 1 if self.packet_tracing else 0))
            if self.hide_emitters:
                check(lib().har_integrator_set_hide_emitters(h, 1))
            if self.samples_per_pass is not None:
                check(lib().har_integrator_set_samples_per_pass(h, self.samples_per_pass))
        return self._h

    def pass_layout(self, sensor, spp=0):
        """(spp_per_pass, n_passes) of a render() with `spp` samples (integrator.cpp:173-183,276-294)"""
        spp = spp or sensor.sampler().sample_count()
        a = C.c_uint32(); b = C.c_uint32()
        check(lib().har_render_pass_layout(self._handle(), C.byref(sensor.har), spp, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __del__(self):
        try:
            if self._h is not None:
                lib().har_integrator_destroy(self._h)
        except Exception:
            pass

    def set_profiling(self, enable):
        check(lib().har_integrator_set_profiling(self._handle(), 1 if enable else 0))

    def timing(self):
        """average per frame since set_profiling(True): {class: (milliseconds, launches)}; "frames" = (frames averaged, 0)"""
        ms = (C.c_float * 8)(); cnt = (C.c_uint32 * 8)()
        check(lib().har_render_timing(self._handle(), ms, cnt))
        names = ["raygen", "trace_closest", "shade", "resolve", "splat", "total", "other", "frames"]
        return {names[i]: (ms[i], cnt[i]) for i in range(8)}

    def stats(self):
        st = _capi.HarStats()
        check(lib().har_render_stats(self._handle(), C.byref(st)))
        return dict(paths=st.paths, vertices=st.vertices, closest_rays=st.closest_rays, shadow_rays=st.shadow_rays)

    def _sensor(self, scene, sensor):
        return scene.sensors()[sensor] if isinstance(sensor, int) else sensor

    def render_film(self, scene, sensor=0, seed=0, spp=0, lanes=None, film=None, alpha_film=None, film_window=None):
        """Raw {R,G,B,W} accumulation of lanes [begin, end) (all when None); no develop.  `alpha_film` (H x W x 4 zeros): also accumulate
        w * alpha into its channel 3 (`rgba` films, har_integrator_set_alpha_film).  `film_window` = (first row, rows): `film` / `alpha_film` hold only those rows
        of the crop window (har_integrator_set_film_window: a rank's band + the filter's reach)."""
        torch = _torch(); dev = _device()
        sensor = self._sensor(scene, sensor)
        if spp:
            sensor.sampler().set_sample_count(spp)
        spp = sensor.sampler().sample_count()
        w, h = sensor.film().crop_size()
        if film is None:
            film = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        lb, le = lanes if lanes else (0, 0)
        check(lib().har_integrator_set_alpha_film(self._handle(), _ptr(alpha_film) if alpha_film is not None else None))
        if film_window is not None:
            if film.shape[0] < film_window[1] or (alpha_film is not None and alpha_film.shape[0] < film_window[1]):
                raise RuntimeError("render_film(): the film holds %d rows, the window %d" % (film.shape[0], film_window[1]))
            check(lib().har_integrator_set_film_window(self._handle(), int(film_window[0]), int(film_window[1])))
        try:
            check(lib().har_render(scene._handle(), self._handle(), C.byref(sensor.har), (sensor.sampler().m_base_seed + int(seed)) & 0xffffffff,
                                   spp, lb, le, _ptr(film), _stream()))
        finally:
            if alpha_film is not None:        # the integrator must not keep a pointer into a tensor it does not own
                check(lib().har_integrator_set_alpha_film(self._handle(), None))
            if film_window is not None:
                check(lib().har_integrator_set_film_window(self._handle(), 0, 0))
        return film

    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True):
        """SamplingIntegrator::render (integrator.cpp:151) / ADIntegrator.render (common.py:46)."""
        torch = _torch()
        sensor = self._sensor(scene, sensor)
        alpha = None
        if sensor.film().alpha and develop:
            w, h = sensor.film().crop_size()
            alpha = torch.zeros((h, w, 4), dtype=torch.float32, device=_device())
        film = self.render_film(scene, sensor, seed, spp, alpha_film=alpha)
        if not develop:
            if self.type == 'prb':
                raise Exception("develop=True must be specified when invoking AD integrators")
            out = film
        else:
            out = develop_film(film, alpha, sensor.film().colour)
        if evaluate:
            torch.cuda.current_stream().synchronize()
            self.stats()                 # surfaces device-side errors (traversal stack overflow)
        return out

    def sample(self, scene, sampler, ray, medium=None, active=True, seed=None):
        """SamplingIntegrator::sample (integrator.h:432-437): (spec 3 x n, valid n) for the rays `ray` (mi.Ray3f) with the sampler's streams.
        `sampler` is a seeded mi.Sampler whose wavefront size equals the number of rays; its state is advanced (`path`) as the call draws from it."""
        torch = _torch(); dev = _device()
        if medium is not None:
            raise RuntimeError("Integrator.sample(): participating media are not part of the hip_ad_rgb path")
        n = len(ray)
        if not sampler.seeded() or sampler.wavefront_size() != n:
            raise RuntimeError("Integrator.sample(): the sampler must be seeded with wavefront_size == number of rays")
        rgb = torch.empty((3, n), dtype=torch.float32, device=dev); valid = torch.empty(n, dtype=torch.uint8, device=dev)
        state_out = torch.empty_like(sampler.state) if self.type == 'path' else None
        sd = sampler.m_seed_value if seed is None else (sampler.m_base_seed + int(seed)) & 0xffffffff
        # `active`: a masked ray never enters the loop -- zero radiance, invalid, and its sampler stream is not advanced (handled inside the call)
        # the mask tensor must OUTLIVE the call: a temporary would go back to the allocator pool the moment its pointer is taken, and the workspace the call
        # allocates through the same pool on first use could be handed its block (round 4's latent fault: the first masked sample() of an integrator read a
        # mask that the workspace's counters had overwritten)
        mask = _mask(active, n)
        check(lib().har_integrator_sample(scene._handle(), self._handle(), sd, 0, n, _ptr(ray.o), _ptr(ray.d), _ptr(ray.maxt), _ptr(sampler.state),
                                          _ptr(mask), _ptr(rgb), _ptr(valid), _ptr(state_out), _stream()))
        del mask
        if state_out is not None:
            sampler.state = state_out
        return rgb, valid.to(torch.bool)

    def render_weights(self, scene, sensor=0, seed=0, spp=0, lanes=None):
        """the weight-only splat of RBIntegrator.render_backward (common.py:716-746: a dummy L = 1 film) for lanes [begin, end) (all when None):
        raw H x W x 4 film whose channel 3 is the accumulated filter weight W[px]"""
        torch = _torch(); dev = _device()
        sensor = self._sensor(scene, sensor)
        if spp:
            sensor.sampler().set_sample_count(spp)
        spp = sensor.sampler().sample_count()
        w, h = sensor.film().crop_size()
        film = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        lb, le = lanes if lanes else (0, 0)
        check(lib().har_render_weights(C.byref(sensor.har), (sensor.sampler().m_base_seed + int(seed)) & 0xffffffff, spp, lb, le, _ptr(film), _stream()))
        return film

    def render_backward(self, scene, params, grad_in, sensor=0, seed=0, spp=0, lanes=None, weight_film=None):
        """RBIntegrator.render_backward (common.py:625-783): returns {key: gradient tensor}."""
        if self.type != 'prb':
            raise RuntimeError("render_backward(): only the `prb` integrator implements the adjoint pass in hip_ad_rgb")
        torch = _torch(); dev = _device()
        sensor = self._sensor(scene, sensor)
        if spp:
            sensor.sampler().set_sample_count(spp)
        spp = sensor.sampler().sample_count()
        w, h = sensor.film().crop_size()
        sd = (sensor.sampler().m_base_seed + int(seed)) & 0xffffffff
        lb, le = lanes if lanes else (0, 0)
        if weight_film is None:
            weight_film = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
            check(lib().har_render_weights(C.byref(sensor.har), sd, spp, lb, le, _ptr(weight_film), _stream()))
        grad_in = torch.as_tensor(grad_in, dtype=torch.float32, device=dev)
        colour = sensor.film().colour
        nc = 1 if colour == 1 else 3
        if sensor.film().alpha and grad_in.numel() == h * w * (nc + 1):
            grad_in = grad_in.reshape(h, w, nc + 1)[:, :, :nc]           # alpha is a mask of detached decisions: no gradient
        grad_in = grad_in.reshape(h, w, nc)
        if colour:            # adjoint of develop's colour transform (linear: rows of luminance() / srgb_to_xyz()): d loss / d rgb = M^T d loss / d colour
            m = torch.tensor(_COLOUR_ROWS[colour], dtype=torch.float32, device=dev)
            grad_in = grad_in @ m
        grad_in = grad_in.contiguous()
        g_refl = torch.zeros((len(scene.bsdfs), 3), dtype=torch.float32, device=dev)
        g_tex = [torch.zeros(tuple(t.shape), dtype=torch.float32, device=dev) for t in scene.textures]
        ptrs = (C.c_void_p * max(1, len(g_tex)))(*[t.data_ptr() for t in g_tex])
        g_emit = torch.zeros((max(1, len(scene.emitters)), 3), dtype=torch.float32, device=dev)
        check(lib().har_integrator_set_grad_emitters(self._handle(), _ptr(g_emit) if self.emitter_gradients else None))
        g_pos = {}; g_inst = None; inst_wanted = {}; rect_wanted = {}
        if self.shape_gradients:
            keys = scene._position_keys(); ikeys = scene._instance_keys()
            if self.shape_gradients is True:         # "every eligible mesh / instance": the meshes the adjoint can differentiate, instances if no instanced mesh is purely specular
                wanted = scene._differentiable_position_keys()
                inst_wanted = ikeys if all(scene._bsdf_has_smooth_lobe(m["bsdf"]) for m in scene.meshes[scene.top_mesh_count:]) else {}
            else:
                inst_wanted = {k: ikeys[k] for k in self.shape_gradients if k in ikeys}
                # a rectangle's 'to_world' (rectangle.cpp:199: Differentiable): its four vertices ARE to_world * (+-1, +-1, 0), so the matrix gradient is the chain rule over
                # the vertex-position gradients (rect_wanted: key -> (mesh, its positions key, whether the caller asked for the positions too))
                rkeys = scene._rect_keys()
                for k in self.shape_gradients:
                    if k in rkeys:
                        pk = scene.meshes[rkeys[k]]["key"] + ".positions"
                        if scene.meshes[rkeys[k]]["emitter"] >= 0 or pk not in scene._differentiable_position_keys():
                            raise RuntimeError("%s: only a rectangle without an area light and with a non-delta BSDF lobe can be differentiated through its 'to_world' in hip_ad_rgb" % k)
                        rect_wanted[k] = (rkeys[k], pk, pk in self.shape_gradients)
                wanted = {k: keys[k] for k in list(self.shape_gradients) + [v[1] for v in rect_wanted.values()] if k not in ikeys and k not in rkeys}    # KeyError: not a differentiable mesh / instance
            g_pos = {k: torch.zeros((scene.meshes[m]["V"].shape[0], 3), dtype=torch.float32, device=dev) for k, m in wanted.items()}      # the parameter's shape: N x 3
            by_mesh = {wanted[k]: g for k, g in g_pos.items()}
            pp = (C.c_void_p * max(1, len(scene.meshes)))(*[by_mesh[m].data_ptr() if m in by_mesh else None for m in range(len(scene.meshes))])
            # what the previous call left on must not veto this call's selection (nested vertex positions and instance transforms exclude each other): instances
            # that are no longer wanted go first, then the positions REPLACE the previous selection.  An unchanged selection changes nothing in the library, so an
            # optimisation loop keeps its adjoint workspace from step to step (both setters only rebuild it when the offset table / instance count differs).
            if not inst_wanted:
                check(lib().har_integrator_set_grad_instances(self._handle(), None, None))
            elif any(m >= scene.top_mesh_count for m in by_mesh):
                raise RuntimeError("Cannot differentiate instance parameters and shapegroup internal parameters at the same time!")     # instance.cpp:162-166
            check(lib().har_integrator_set_grad_positions(self._handle(), scene._handle(), pp if g_pos else None))
            if inst_wanted:
                g_inst = torch.zeros((len(scene.instances), 12), dtype=torch.float32, device=dev)
            check(lib().har_integrator_set_grad_instances(self._handle(), scene._handle() if inst_wanted else None, _ptr(g_inst) if inst_wanted else None))
        else:
            check(lib().har_integrator_set_grad_positions(self._handle(), None, None))
            check(lib().har_integrator_set_grad_instances(self._handle(), None, None))
        g_extra = torch.zeros((max(1, len(scene.bsdfs)), 15), dtype=torch.float32, device=dev) if self.bsdf_parameter_gradients else None
        check(lib().har_integrator_set_grad_bsdf_params(self._handle(), _ptr(g_extra) if g_extra is not None else None))
        check(lib().har_integrator_set_grad_light_texels(self._handle(), 1 if self.light_texel_gradients else 0))
        check(lib().har_render_backward(scene._handle(), self._handle(), C.byref(sensor.har), _ptr(grad_in), _ptr(weight_film), sd, spp,
                                        lb, le, _ptr(g_refl), ptrs, _stream()))
        out = scene._gradients(g_refl, g_tex, g_emit if self.emitter_gradients else None)
        if g_extra is not None:
            for k, (what, b) in scene._bsdf_param_keys().items():
                if what == "ior":
                    continue                  # (no gradient w.r.t. the scalar index of refraction)
                rec = g_extra[b.index]
                out[k] = {"alpha": rec[0:6].sum().reshape(1), "alpha_u": rec[0:3].sum().reshape(1), "alpha_v": rec[3:6].sum().reshape(1),
                          "eta": rec[6:9], "k": rec[9:12], "slot1": rec[12:15]}[what]
        if self.light_texel_gradients:            # the light's bitmap is a texture of the scene like any other: its gradient sits in its entry of grad_textures
            for k, (kind, i) in scene._pose_keys().items():
                if kind == "emitter_tex":
                    out[k] = g_tex[scene.emitters[i]["light"].tex_index]
        out.update(g_pos)
        for k, (mi_, pk, keep) in rect_wanted.items():         # d loss / d to_world[r, :] = sum over the vertices of  d loss / d p_v[r] * (local corner of v, 1);  fourth row constant
            rect = scene.meshes[mi_]["rect"]
            M = np.asarray(rect["to_world"].matrix, np.float64).reshape(4, 4)
            Pw = np.concatenate([scene.meshes[mi_]["V"][:, :3].astype(np.float64), np.ones((4, 1))], axis=1)
            local = torch.as_tensor((np.linalg.inv(M) @ Pw.T).T, dtype=torch.float32, device=dev)              # 4 vertices x (x, y, 0, 1)
            g = torch.zeros((4, 4), dtype=torch.float32, device=dev); g[:3, :] = g_pos[pk].reshape(4, 3).T @ local
            out[k] = g
            if not keep:
                out.pop(pk, None)
        for k, i in inst_wanted.items():                      # column-major 3x4 -> the reference's 4x4 (constant fourth row: zero gradient)
            m = torch.zeros((4, 4), dtype=torch.float32, device=dev); m[:3, :] = g_inst[i].reshape(4, 3).T
            out[k] = m
        return out


def _render_forward(self, scene, params=None, sensor=0, seed=0, spp=0, tangents=None, lanes=None, develop=True):
    """RBIntegrator.render_forward (common.py:497-623): the forward-mode derivative image for the parameter tangents.
    `tangents` = {key: tensor} (the values dr.set_grad() would carry); without it, the `.grad` fields of the `params` entries that have
    requires_grad are used.  Keys are those of mi.traverse(scene); missing keys have a zero tangent.  Returns the gradient image H x W x 3."""
    if self.type != 'prb':
        raise RuntimeError("render_forward(): only the `prb` integrator implements the differential pass in hip_ad_rgb")
    torch = _torch(); dev = _device()
    sensor = self._sensor(scene, sensor)
    if spp:
        sensor.sampler().set_sample_count(spp)
    spp = sensor.sampler().sample_count()
    w, h = sensor.film().crop_size()
    if tangents is None:
        tangents = {k: v.grad for k, v in (params or {}).items() if getattr(v, 'requires_grad', False) and getattr(v, 'grad', None) is not None}
    keys = scene._param_keys()
    unknown = [k for k in tangents if k not in keys]
    if unknown:
        raise RuntimeError("render_forward(): %s are not differentiable parameters of the `prb` forward mode (available: %s)" % (unknown, sorted(keys)))
    t_refl = torch.zeros((max(1, len(scene.bsdfs)), 3), dtype=torch.float32, device=dev)
    t_tex = [torch.zeros(tuple(t.shape), dtype=torch.float32, device=dev) for t in scene.textures]
    t_emit = torch.zeros((max(1, len(scene.emitters)), 3), dtype=torch.float32, device=dev)
    any_emit = False
    for k, v in tangents.items():
        kind, b = keys[k]
        v = torch.as_tensor(v, dtype=torch.float32, device=dev)
        if kind == "tex":
            t_tex[b.tex_index].copy_(v.reshape(t_tex[b.tex_index].shape))
        elif kind == "emit":
            t_emit[b].copy_(v.reshape(3)); any_emit = True
        else:
            t_refl[b.index].copy_(v.reshape(3))
    ptrs = (C.c_void_p * max(1, len(t_tex)))(*[t.data_ptr() for t in t_tex])
    film = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    lb, le = lanes if lanes else (0, 0)
    check(lib().har_render_forward(scene._handle(), self._handle(), C.byref(sensor.har), (sensor.sampler().m_base_seed + int(seed)) & 0xffffffff, spp, lb, le,
                                   _ptr(t_refl), ptrs if t_tex else None, _ptr(t_emit) if any_emit else None, _ptr(film), _stream()))
    return develop_film(film, None, sensor.film().colour) if develop else film


Integrator.render_forward = _render_forward


class Bitmap:
    """Bitmap (src/core/bitmap.cpp) restricted to what HDRFilm::write needs: float32 H x W x {1,3,4}, write() to .exr / .pfm"""

    def __init__(self, array):
        if isinstance(array, (str, os.PathLike)):             # Bitmap(filename): OpenEXR (NO / ZIPS / ZIP) and PFM, read by the C++ host library
            img = _capi.HarImage()
            check(lib().har_image_read(str(array).encode(), C.byref(img)))
            try:
                array = np.ctypeslib.as_array(img.data, shape=(img.height, img.width, img.channels)).copy()
            finally:
                lib().har_image_free(C.byref(img))
        if hasattr(array, 'detach'):
            array = array.detach().cpu().numpy()
        a = _f32(array)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.ndim != 3:
            raise RuntimeError("Bitmap: expected an H x W x C array")
        self.data = np.ascontiguousarray(a)

    def size(self):
        return (self.data.shape[1], self.data.shape[0])

    def channel_count(self):
        return self.data.shape[2]

    def write(self, filename):
        h, w, c = self.data.shape
        ext = str(filename).lower().rsplit('.', 1)[-1]
        if ext == 'exr':
            check(lib().har_image_write_exr(str(filename).encode(), _fp(self.data), w, h, c))
        elif ext == 'pfm':
            check(lib().har_image_write_pfm(str(filename).encode(), _fp(self.data), w, h, c))
        else:
            raise RuntimeError("Bitmap.write(): unsupported file format \"%s\" (exr and pfm are implemented)" % ext)


def write_bitmap(filename, data):
    """mi.util.write_bitmap (src/python/python/util.py)"""
    Bitmap(data).write(filename)


# luminance() and srgb_to_xyz() (include/mitsuba/core/spectrum.h:439-442, 402-410): rows of the colour transform HDRFilm::develop applies for `luminance*` / `xyz*` films
_COLOUR_ROWS = {0: None, 1: ((0.212671, 0.715160, 0.072169),),
                2: ((0.412453, 0.357580, 0.180423), (0.212671, 0.715160, 0.072169), (0.019334, 0.119193, 0.950227))}


def develop_film(film, alpha_film=None, colour=0):
    """HDRFilm::develop (hdrfilm.cpp:301-404): colour / W -- RGB, Y = luminance(rgb) or XYZ = srgb_to_xyz(rgb) by the film's pixel format -- and A / W for films with
    an alpha channel (alpha_film: channel 3 holds the accumulated w * alpha)"""
    torch = _torch()
    h, w, _ = film.shape
    img = torch.empty((h, w, 1 if colour == 1 else 3), dtype=torch.float32, device=film.device)
    check(lib().har_film_develop_format(_ptr(film), w, h, int(colour), _ptr(img), _stream()))
    if alpha_film is None:
        return img
    weight = film[:, :, 3:4]
    a = alpha_film[:, :, 3:4] / torch.where(weight == 0, torch.ones_like(weight), weight)
    return torch.cat([img, a], dim=2)


# ---------------------------------------------------------------------------
#  Scene
# ---------------------------------------------------------------------------

def _static_table(fn):
    """the key tables of a scene (which parameters exist, under which names) are fixed at construction: built once (an optimisation loop asks every step)"""
    name = "_memo" + fn.__name__

    def get(self):
        t = self.__dict__.get(name)
        if t is None:
            t = self.__dict__[name] = fn(self)
        return t
    get.__name__ = fn.__name__; get.__doc__ = fn.__doc__
    return get


class Scene:
    """Scene (src/render/scene.cpp): owns the flat SceneIR arrays and the device accel."""

    def __init__(self, children):
        self.bsdf_objs = []; self.meshes = []; self.top_mesh_count = 0
        self.groups = []; self.instances = []; self.instance_keys = []; self.emitters = []
        self.m_sensors = []; self.sensor_keys = []; self.m_integrator = None; self.textures = []; self.texture_modes = []; self.texture_to_uv = []
        self._h = None; self._keep = []
        self._device_values = {}      # (kind, index) -> (device tensor, record): values params.update() pushed device-to-device; the numpy mirrors are refreshed by sync_host()
        named = {}
        shapes = []; groups = []; insts = []
        # Scene::emitters() order = declaration order of the children (scene.cpp:40-70): shapes with an area emitter and
        # stand-alone emitters; the uniform emitter selection of sample_emitter() depends on it
        self._emitter_order = [key for key, obj in children.items()
                               if (isinstance(obj, Mesh) and obj.emitter is not None) or isinstance(obj, (ConstantEmitter, EnvmapEmitter, PointLight, SpotLight, DirectionalEmitter))]
        self.emitters = [None] * len(self._emitter_order)
        if sum(isinstance(o, (ConstantEmitter, EnvmapEmitter)) for o in children.values()) > 1:
            raise RuntimeError("Only one environment emitter can be specified per scene.")
        for key, obj in children.items():
            if isinstance(obj, ConstantEmitter):
                self.emitters[self._emitter_order.index(key)] = dict(type=1, mesh=0xffffffff, radiance=obj.radiance, to_world=[0.0] * 12,
                                                                     normal=[0.0] * 3, inv_area=0.0, sampling_weight=obj.sampling_weight)
            elif isinstance(obj, PointLight):                   # HarEmitter type 4: `radiance` = intensity, to_world[9..11] = position
                self.emitters[self._emitter_order.index(key)] = dict(type=4, mesh=0xffffffff, radiance=obj.intensity,
                                                                     to_world=[1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0] + [float(x) for x in obj.position],
                                                                     normal=[0.0] * 3, inv_area=0.0, sampling_weight=obj.sampling_weight)
            elif isinstance(obj, SpotLight):                    # HarEmitter type 5: transform + inverse, normal = (cutoff_angle, beam_width, -) in degrees
                self.emitters[self._emitter_order.index(key)] = dict(type=5, mesh=0xffffffff, radiance=obj.intensity, to_world=obj.to_world.col_major_3x4(),
                                                                     to_local=obj.to_world.inverse().col_major_3x4(),
                                                                     normal=[obj.cutoff_angle, obj.beam_width, 0.0], inv_area=0.0, sampling_weight=obj.sampling_weight)
            elif isinstance(obj, DirectionalEmitter):           # HarEmitter type 6: `radiance` = irradiance, the transform (light travels along its +z)
                self.emitters[self._emitter_order.index(key)] = dict(type=6, mesh=0xffffffff, radiance=obj.irradiance, to_world=obj.to_world.col_major_3x4(),
                                                                     to_local=obj.to_world.inverse().col_major_3x4(), normal=[0.0] * 3, inv_area=0.0, sampling_weight=obj.sampling_weight)
            elif isinstance(obj, EnvmapEmitter):                # the radiance image travels in the texture table
                self.emitters[self._emitter_order.index(key)] = dict(
                    type=2, mesh=len(self.textures), radiance=[obj.scale, 1.0 if obj.mis_compensation else 0.0, 0.0],
                    to_world=obj.to_world.col_major_3x4(), to_local=obj.to_world.inverse().col_major_3x4(), normal=[0.0] * 3, inv_area=0.0, sampling_weight=obj.sampling_weight)
                self.textures.append(obj.data); self.texture_modes.append(0); self.texture_to_uv.append(None)
        for key, obj in children.items():
            if isinstance(obj, BSDF):
                named[key] = obj
                if obj.id is None:
                    obj.id = key
                self._add_bsdf(obj)
        for key, obj in children.items():
            if isinstance(obj, Sensor):
                self.m_sensors.append(obj); self.sensor_keys.append(key)
            elif isinstance(obj, Integrator):
                self.m_integrator = obj
            elif isinstance(obj, Mesh):
                shapes.append((key, obj))
            elif isinstance(obj, ShapeGroup):
                groups.append((key, obj))
            elif isinstance(obj, Instance):
                insts.append((key, obj))
        for key, m in shapes:
            self._add_mesh(key, m)
        self.top_mesh_count = len(self.meshes)
        gindex = {}
        for key, g in groups:
            first = len(self.meshes)
            for i, m in enumerate(g.shapes):
                if m.emitter is not None:
                    raise RuntimeError("Instancing of emitters is not supported")
                self._add_mesh("%s.%s" % (key, g.keys[i]), m)
            gindex[id(g)] = len(self.groups)
            self.groups.append((first, len(self.meshes) - first))
        for key, it in insts:
            if id(it.group) not in gindex:
                raise RuntimeError("A reference to a 'shapegroup' must be specified!")
            self.instances.append((gindex[id(it.group)], it.to_world.col_major_3x4(), it.to_world.inverse().col_major_3x4()))
            self.instance_keys.append(key)

    # -- construction helpers
    def _add_bsdf(self, b):
        if b.scene is self:
            return b.index
        b.scene = self; b.index = len(self.bsdf_objs)
        if b.texture is not None:
            b.tex_index = len(self.textures); self.textures.append(b.texture); self.texture_modes.append(b.tex_mode); self.texture_to_uv.append(b.tex_to_uv)
        self.bsdf_objs.append(b)
        if getattr(b, 'back', None) is not None:
            self._add_bsdf(b.back)
        return b.index

    @property
    def bsdfs(self):
        return self.bsdf_objs

    def _add_mesh(self, key, m):
        if m.bsdf is None:          # Shape(props), shape.cpp:50-57: a default diffuse BSDF -- black when the shape carries an emitter
            m.bsdf = BSDF({'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.0, 0.0, 0.0]}} if m.emitter is not None else None)
        if m.bsdf.scene is None:
            m.bsdf.id = key + ".bsdf"             # a BSDF nested in a shape (not a scene-level object): '<shape>.bsdf.reflectance.value' as in mi.traverse() (Shape::traverse registers it as "bsdf")
        bi = self._add_bsdf(m.bsdf)
        em = -1
        if m.emitter is not None:
            em = self._emitter_order.index(key)
            light = getattr(m, 'emitter_light', None)
            if light is not None and light.texture is not None:
                if not hasattr(m, 'rect'):
                    raise RuntimeError("area: a bitmap `radiance` is implemented on `rectangle` shapes (Rectangle::eval_parameterization, src/shapes/rectangle.cpp:215-237); on a "
                                       "triangle mesh the reference maps the sampled uv through Mesh::eval_parameterization (a ray cast against the mesh in uv space, "
                                       "src/render/mesh.cpp), which hip_ad_rgb does not have")
                light.tex_index = len(self.textures); self.textures.append(light.texture); self.texture_modes.append(light.tex_mode); self.texture_to_uv.append(light.tex_to_uv)
                self.emitters[em] = dict(type=7, mesh=len(self.meshes), radiance=[0.0, 0.0, 0.0], to_world=m.rect['to_world'].col_major_3x4(), normal=m.rect['normal'],
                                         inv_area=m.rect['inv_area'], sampling_weight=getattr(m, 'emitter_weight', 1.0), radiance_texture=light.tex_index, light=light)
            elif hasattr(m, 'rect'):        # Rectangle::sample_position (analytic parameterisation)
                self.emitters[em] = (dict(type=0, mesh=len(self.meshes), radiance=m.emitter, to_world=m.rect['to_world'].col_major_3x4(),
                                          normal=m.rect['normal'], inv_area=m.rect['inv_area'], sampling_weight=getattr(m, 'emitter_weight', 1.0)))
            else:                         # any other triangle mesh: Mesh::sample_position (area-weighted face selection)
                self.emitters[em] = dict(type=3, mesh=len(self.meshes), radiance=m.emitter, to_world=[0.0] * 12, normal=[0.0] * 3, inv_area=0.0,
                                         sampling_weight=getattr(m, 'emitter_weight', 1.0))
        self.meshes.append(dict(key=key, V=np.ascontiguousarray(m.V), F=np.ascontiguousarray(m.F), bsdf=bi, emitter=em, flags=m.flags, rect=getattr(m, 'rect', None)))

    def sync_host(self):
        """refresh the host mirrors (Scene.textures, BSDF.value / .texture, emitters[i]['radiance']) from the values params.update() pushed device-to-device"""
        for (kind, idx), (t, b) in list(self._device_values.items()):
            v = t.detach().to("cpu").numpy().astype(np.float32)
            if kind == "emit":
                self.emitters[idx]["radiance"] = np.ascontiguousarray(v.reshape(3))
            elif kind == "tex":
                b.texture = np.ascontiguousarray(v.reshape(b.texture.shape)); self.textures[idx] = b.texture
            else:
                b.value = np.ascontiguousarray(v.reshape(3))
        self._device_values.clear()
        self._sync_host_geometry()

    def _sync_host_geometry(self):
        """the packed vertex records of the meshes whose positions were updated ON THE DEVICE (har_scene_update_vertices_device): positions and regenerated normals
        back into meshes[i]['V'] -- needed before anything rebuilds the scene from the host mirrors or reads them"""
        stale = getattr(self, "_stale_meshes", None)
        if stale:
            if self._h is not None:
                for i in sorted(stale):
                    V = self.meshes[i]["V"] = np.ascontiguousarray(self.meshes[i]["V"], np.float32)
                    check(lib().har_scene_get_vertices(self._h, int(i), _fp(V), _stream()))
            stale.clear()
        if getattr(self, "_stale_instances", False):       # to_world / to_object of the instances were updated on the device (har_scene_update_instances_device)
            if self._h is not None and self.instances:
                n = len(self.instances)
                tw = np.zeros((n, 12), np.float32); to = np.zeros((n, 12), np.float32)
                check(lib().har_scene_get_instances(self._h, 0, n, _fp(tw), _fp(to), _stream()))
                self.instances = [(self.instances[i][0], [float(x) for x in tw[i]], [float(x) for x in to[i]]) for i in range(n)]
            self._stale_instances = False

    def _drop_handle(self, keep_geometry=True):
        """the next render builds a new scene handle from the host mirrors: bring them up to date first (unless the handle is being dropped BECAUSE an update failed)"""
        if self._h is None:
            return
        if keep_geometry:
            self._sync_host_geometry()
        else:
            getattr(self, "_stale_meshes", set()).clear(); self._stale_instances = False
        lib().har_scene_destroy(self._h); self._h = None

    # -- C ABI description
    def desc(self):
        self.sync_host()
        M = _capi
        meshes = (M.HarMesh * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            meshes[i].vertex_ptr = _fp(m["V"]); meshes[i].index_ptr = _up(m["F"])
            meshes[i].vertex_count = m["V"].shape[0]; meshes[i].face_count = m["F"].shape[0]
            meshes[i].bsdf = m["bsdf"]; meshes[i].emitter = m["emitter"]; meshes[i].flags = m["flags"]
        groups = (M.HarShapeGroup * max(1, len(self.groups)))()
        for i, (a, b) in enumerate(self.groups):
            groups[i].first_mesh = a; groups[i].mesh_count = b
        insts = (M.HarInstance * max(1, len(self.instances)))()
        for i, (g, tw, to) in enumerate(self.instances):
            insts[i].group = g
            insts[i].to_world = (C.c_float * 12)(*[float(x) for x in tw]); insts[i].to_object = (C.c_float * 12)(*[float(x) for x in to])
        bsdfs = (M.HarBSDF * max(1, len(self.bsdf_objs)))()
        for i, b in enumerate(self.bsdf_objs):
            bsdfs[i].type = BSDF_TYPES[b.kind]; bsdfs[i].texture = b.tex_index if b.texture is not None else -1
            bsdfs[i].reflectance = (C.c_float * 3)(*[float(x) for x in b.value])
            bsdfs[i].flags = b.flags; bsdfs[i].reflectance2 = (C.c_float * 3)(*[float(x) for x in b.value2])
            bsdfs[i].alpha_u = b.alpha_u; bsdfs[i].alpha_v = b.alpha_v; bsdfs[i].eta = b.eta
            bsdfs[i].eta_c = (C.c_float * 3)(*[float(x) for x in b.eta_c]); bsdfs[i].k_c = (C.c_float * 3)(*[float(x) for x in b.k_c])
            bsdfs[i].back = b.back.index if b.back is not None else -1
        texs = (M.HarTexture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            texs[i].data = _fp(t); texs[i].height = t.shape[0]; texs[i].width = t.shape[1]; texs[i].mode = self.texture_modes[i] if i < len(self.texture_modes) else 0
            if i < len(self.texture_to_uv) and self.texture_to_uv[i] is not None:
                texs[i].to_uv = (C.c_float * 6)(*self.texture_to_uv[i])
        ems = (M.HarEmitter * max(1, len(self.emitters)))()
        for i, e in enumerate(self.emitters):
            self._fill_emitter_record(ems[i], e)
        d = M.HarSceneDesc()
        d.meshes = meshes; d.mesh_count = len(self.meshes); d.top_mesh_count = self.top_mesh_count
        d.groups = groups; d.group_count = len(self.groups)
        d.instances = insts; d.instance_count = len(self.instances)
        d.bsdfs = bsdfs; d.bsdf_count = len(self.bsdf_objs)
        d.textures = texs; d.texture_count = len(self.textures)
        d.emitters = ems; d.emitter_count = len(self.emitters)
        self._keep = [meshes, groups, insts, bsdfs, texs, ems]
        return d

    @staticmethod
    def _fill_emitter_record(rec, e):
        rec.type = e.get("type", 0); rec.mesh = e["mesh"]
        rec.radiance = (C.c_float * 3)(*[float(x) for x in e["radiance"]])
        rec.to_world = (C.c_float * 12)(*[float(x) for x in e["to_world"]])
        rec.normal = (C.c_float * 3)(*[float(x) for x in e["normal"]]); rec.inv_area = float(e["inv_area"])
        rec.to_local = (C.c_float * 12)(*[float(x) for x in e.get("to_local", [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0])])
        rec.sampling_weight = float(e.get("sampling_weight", 1.0)); rec.radiance_texture = int(e.get("radiance_texture", 0))

    def _handle(self):
        if self._h is None:
            _device()
            d = self.desc(); h = C.c_void_p()
            check(lib().har_scene_create(C.byref(d), C.byref(h)))
            self._h = h
        return self._h

    def __del__(self):
        try:
            if self._h is not None:
                lib().har_scene_destroy(self._h)
        except Exception:
            pass

    # -- reference API
    def sensors(self):
        return self.m_sensors

    def integrator(self):
        return self.m_integrator

    def shapes(self):
        return self.meshes

    def accel_info(self):
        info = (C.c_uint64 * 4)()
        check(lib().har_scene_accel_info(self._handle(), info))
        return dict(nodes=info[0], triangles=info[1], bytes=info[2], depth=info[3])

    def _intersect(self, ray, naive, active=True):
        torch = _torch(); dev = _device(); n = len(ray)
        t = torch.empty(n, dtype=torch.float32, device=dev); u = torch.empty_like(t); v = torch.empty_like(t)
        prim = torch.empty(n, dtype=torch.int32, device=dev); shape = torch.empty_like(prim); inst = torch.empty_like(prim)
        mask = _mask(active, n)          # kept alive across the call (see Integrator.sample)
        check(lib().har_ray_intersect_preliminary(self._handle(), n, _ptr(ray.o), _ptr(ray.d), _ptr(ray.maxt), _ptr(mask), 1 if naive else 0,
                                                  _ptr(t), _ptr(u), _ptr(v), _ptr(prim), _ptr(shape), _ptr(inst), _stream()))
        return PreliminaryIntersection3f(self, t, u, v, prim, shape, inst)

    def ray_intersect_preliminary(self, ray, coherent=False, reorder=False, reorder_hint=0, reorder_hint_bits=0, active=True):
        """Scene::ray_intersect_preliminary (scene.cpp:216-230).  coherent / reorder* are scheduling hints of the reference's backends (DRJIT_MARK_USED)."""
        return self._intersect(ray, False, active)

    def _intersect_si(self, ray, ray_flags, active, naive):
        torch = _torch(); dev = _device(); n = len(ray)
        t = torch.empty(n, dtype=torch.float32, device=dev); u = torch.empty_like(t); v = torch.empty_like(t)
        prim = torch.empty(n, dtype=torch.int32, device=dev); shape = torch.empty_like(prim); inst = torch.empty_like(prim)
        out = torch.empty((33, n), dtype=torch.float32, device=dev)
        flags = RayFlags.Default if ray_flags is None else int(ray_flags)
        mask = _mask(active, n)          # kept alive across the call (see Integrator.sample)
        check(lib().har_ray_intersect(self._handle(), n, _ptr(ray.o), _ptr(ray.d), _ptr(ray.maxt), flags, _ptr(mask), 1 if naive else 0,
                                      _ptr(t), _ptr(u), _ptr(v), _ptr(prim), _ptr(shape), _ptr(inst), _ptr(out), _stream()))
        si = SurfaceInteraction3f(out)
        si.prim_index = prim; si.shape_index = shape; si.instance = inst
        return si

    def ray_intersect(self, ray, ray_flags=None, coherent=False, reorder=False, reorder_hint=0, reorder_hint_bits=0, active=True):
        """Scene::ray_intersect(ray, ray_flags, coherent, ..., active) (scene.cpp:197-214)"""
        return self._intersect_si(ray, ray_flags, active, False)

    def ray_intersect_naive(self, ray, active=True):
        return self._intersect_si(ray, None, active, True)

    def ray_test(self, ray, coherent=False, active=True, naive=False):
        torch = _torch(); dev = _device(); n = len(ray)
        hit = torch.empty(n, dtype=torch.uint8, device=dev)
        mask = _mask(active, n)          # kept alive across the call (see Integrator.sample)
        check(lib().har_ray_test(self._handle(), n, _ptr(ray.o), _ptr(ray.d), _ptr(ray.maxt), _ptr(mask), 1 if naive else 0, _ptr(hit), _stream()))
        return hit.bool()

    def _compute_si(self, ray, pi, ray_flags=1, active=True):
        torch = _torch(); dev = _device(); n = len(ray)
        out = torch.empty((33, n), dtype=torch.float32, device=dev)
        mask = _mask(active, n)          # kept alive across the call (see Integrator.sample)
        check(lib().har_compute_surface_interaction(self._handle(), n, _ptr(ray.o), _ptr(ray.d), _ptr(pi.t), _ptr(pi.prim_uv[0]), _ptr(pi.prim_uv[1]),
                                                    _ptr(pi.prim_index), _ptr(pi.shape_index), _ptr(pi.instance), int(ray_flags), _ptr(mask),
                                                    _ptr(out), _stream()))
        return SurfaceInteraction3f(out)

    # -- parameters (mi.traverse)
    @staticmethod
    def _bsdf_key_base(b):
        """the prefix of a BSDF's parameters in mi.traverse(): its id -- the scene-level key, or '<shape>.bsdf' -- and, for the BSDFs nested in a `twosided`, the names
        TwoSidedBRDF::traverse registers them under (twosided.cpp:106-109): '<twosided>.brdf_0' and, when the back side is a BSDF of its own, '<twosided>.brdf_1'
        (util.py:320-334 walks an object once, so a twosided with ONE nested BSDF has no brdf_1 entries)"""
        parent = getattr(b, 'twosided_parent', None)
        if parent is not None:
            return (parent.id if parent.id else "bsdf%d" % parent.index) + ".brdf_1"
        base = b.id if b.id else "bsdf%d" % b.index
        return base + ".brdf_0" if (b.flags & 1) else base

    @_static_table
    def _param_keys(self):
        keys = {}
        for b in self.bsdf_objs:
            base = self._bsdf_key_base(b)
            if b.texture is not None:
                keys[base + "." + b.slot0_name + ".data"] = ("tex", b)
            else:
                keys[base + "." + b.slot0_name + ".value"] = ("rgb", b)
        # emitter radiances: `<shape>.emitter.radiance.value` for area lights, `<emitter>.radiance.value` for `constant` (type 2 = envmap: none)
        for i, key in enumerate(self._emitter_order):
            e = self.emitters[i]
            if e.get("type", 0) in (0, 3):
                keys[key + ".emitter.radiance.value"] = ("emit", i)
            elif e["type"] == 1:
                keys[key + ".radiance.value"] = ("emit", i)
            elif e["type"] in (4, 5):                             # PointLight::traverse (point.cpp:84-88), SpotLight::traverse (spot.cpp:111-118): `intensity`
                keys[key + ".intensity.value"] = ("emit", i)
            elif e["type"] == 6:                                  # DirectionalEmitter::traverse (directional.cpp:93-97)
                keys[key + ".irradiance.value"] = ("emit", i)
        return keys

    @_static_table
    def _bsdf_param_keys(self):
        """the non-slot-0 parameters of the rough models: '<bsdf>.alpha.value' (or alpha_u / alpha_v), '<bsdf>.eta.value', '<bsdf>.k.value' of
        roughconductor (roughconductor.cpp:212-225 traverse), '<bsdf>.alpha' and '<bsdf>.specular_reflectance.value' of roughplastic (roughplastic.cpp:208-215)"""
        keys = {}
        for b in self.bsdf_objs:
            base = self._bsdf_key_base(b)
            if b.kind == 'roughplastic':
                keys[base + ".alpha"] = ("alpha", b)                  # a Float member, not a texture: no '.value' (roughplastic.cpp:210, :534)
            elif b.kind == 'roughconductor':
                if b.anisotropic:
                    keys[base + ".alpha_u.value"] = ("alpha_u", b); keys[base + ".alpha_v.value"] = ("alpha_v", b)
                else:
                    keys[base + ".alpha.value"] = ("alpha", b)
            if b.kind == 'roughconductor':
                keys[base + ".eta.value"] = ("eta", b); keys[base + ".k.value"] = ("k", b)
            if b.kind == 'roughplastic' and b.slot1_name:
                keys[base + "." + b.slot1_name + ".value"] = ("slot1", b)
            if b.kind in ('dielectric', 'plastic', 'roughplastic'):
                # the relative index of refraction int_ior / ext_ior, a plain float: dielectric.cpp:238, plastic.cpp:185 (NonDifferentiable), roughplastic.cpp:211 (Differentiable
                # there; updatable without a gradient here)
                keys[base + ".eta"] = ("ior", b)
        return keys

    def _bsdf_param_value(self, what, b):
        return {"alpha": [b.alpha_u], "alpha_u": [b.alpha_u], "alpha_v": [b.alpha_v], "eta": b.eta_c, "k": b.k_c, "slot1": b.value2, "ior": [b.eta]}[what]

    def _set_bsdf_param(self, what, b, v):
        """params.update() of a non-slot-0 BSDF parameter: the record is re-lowered IN PLACE when a scene handle exists (har_scene_set_bsdf_params), else with the first handle (roughplastic's sampling weights and
        transmittance tables depend on alpha / the colours: RoughPlastic::parameters_changed, roughplastic.cpp:204-242)"""
        v = np.asarray(v, np.float32).reshape(-1)
        if what == "ior":              # Dielectric / SmoothPlastic / RoughPlastic::parameters_changed: Fresnel terms, plastic's internal reflectance and tables follow eta --
            if not (float(v[0]) > 0.0) or not math.isfinite(float(v[0])) or (b.kind == 'roughplastic' and float(v[0]) == 1.0):      # all re-derived when the next scene handle lowers the record
                raise RuntimeError("The interior and exterior indices of refraction must be positive" + (" and differ!" if b.kind == 'roughplastic' else "!"))
            b.eta = float(v[0])
            self._drop_handle()
            return
        if what == "alpha":
            b.alpha_u = b.alpha_v = float(v[0])
        elif what == "alpha_u":
            b.alpha_u = float(v[0])
        elif what == "alpha_v":
            b.alpha_v = float(v[0])
        elif what == "eta":
            b.eta_c = _f32(v[:3])
        elif what == "k":
            b.k_c = _f32(v[:3])
        else:
            b.value2 = _f32(v[:3])
        if self._h is not None:
            # the record is re-lowered IN PLACE (har_scene_set_bsdf_params: alpha / eta / k / slot 1 + roughplastic's table, internal reflectance, sampling weight); the scene
            # handle and the acceleration data survive -- rounds 3-5 rebuilt the whole scene for every step of a roughness optimisation
            rec = _capi.HarBSDF()
            rec.type = BSDF_TYPES[b.kind]; rec.flags = b.flags
            rec.reflectance2 = (C.c_float * 3)(*[float(x) for x in b.value2])
            rec.alpha_u = b.alpha_u; rec.alpha_v = b.alpha_v; rec.eta = b.eta
            rec.eta_c = (C.c_float * 3)(*[float(x) for x in b.eta_c]); rec.k_c = (C.c_float * 3)(*[float(x) for x in b.k_c])
            rc = lib().har_scene_set_bsdf_params(self._h, int(b.index), C.byref(rec))
            if rc != 0:
                msg = (lib().har_last_error() or b"").decode()
                self._drop_handle()
                raise RuntimeError(msg or "BSDF parameter update failed")

    @_static_table
    def _position_keys(self):
        """'<shape>.positions' (an N x 3 tensor: Mesh::traverse, src/render/mesh.cpp:827; a flat array of 3 N floats is accepted on write) of the top-level meshes and, as '<group>.<child>.positions', of the
        meshes inside shape groups (object space, shared by all instances); writing them regenerates the vertex normals of a smooth-shaded mesh (mesh.cpp:876-878),
        see _set_vertex_positions"""
        return {m["key"] + ".positions": i for i, m in enumerate(self.meshes) if m["V"].shape[0]}

    def _bsdf_has_smooth_lobe(self, index):
        """BSDFFlags::Smooth on every side: models made of delta lobes only (`dielectric`, `conductor`) cannot sit on MOVING geometry -- their eval() is zero,
        so prb.py:288 would form relative_grad(0) (har_integrator_set_grad_positions)"""
        b = self.bsdf_objs[index]
        delta = ("dielectric", "conductor")
        return b.kind not in delta and (b.back is None or b.back.kind not in delta)

    def _differentiable_position_keys(self):
        """the subset of _position_keys() the `prb` adjoint can differentiate with `shape_gradients=True`: top-level meshes whose BSDF has a non-delta lobe (any of
        diffuse, roughconductor, roughplastic, plastic, plain or inside `twosided`) and, if they carry vertex normals, carry the REGENERATED ones (Mesh::compute_normals:
        what a position update produces, mesh.cpp:876-878) -- a mesh with normals of another origin is left out here and refused when named explicitly; the other
        meshes of the scene may carry any BSDF"""
        out = {}
        for k, i in self._position_keys().items():
            m = self.meshes[i]
            if i >= self.top_mesh_count:         # nested meshes exclude the instances' to_world (instance.cpp:162-166): only when named explicitly
                continue
            if not self._bsdf_has_smooth_lobe(m["bsdf"]):
                continue
            if (m["flags"] & 1) and i not in getattr(self, "_stale_meshes", ()):        # (a mesh updated on the device carries the normals that update regenerated)
                V = np.ascontiguousarray(m["V"]).copy(); F = np.ascontiguousarray(m["F"])
                check(lib().har_mesh_compute_normals(V.shape[0], _fp(V), F.shape[0], _up(F)))
                if np.abs(V[:, 3:6] - m["V"][:, 3:6]).max() > 1e-4:
                    continue
            out[k] = i
        return out

    @_static_table
    def _rect_keys(self):
        """'<rectangle>.to_world' (4 x 4, Rectangle::traverse, src/shapes/rectangle.cpp:197-200 -- the one entry that plugin registers besides its BSDF / emitter) of the
        top-level rectangles"""
        return {m["key"] + ".to_world": i for i, m in enumerate(self.meshes[:self.top_mesh_count]) if m.get("rect") is not None}

    def _rect_matrix(self, i):
        return np.asarray(self.meshes[i]["rect"]["to_world"].matrix, np.float32).reshape(4, 4).copy()

    def _set_rect_to_world(self, i, m4):
        """params['<rectangle>.to_world'] = ...; params.update(): Rectangle::parameters_changed (rectangle.cpp:202-214) re-initialises the shape from the new transform --
        here the four vertex records (positions, normal, uv), the winding, and for a rectangle that carries an area light its sampling record (frame, normal, 1 / area)
        are re-baked by the code that built them (har_shape_rectangle); the next render builds a new scene handle from them"""
        m4 = np.asarray(m4, np.float64).reshape(4, 4)
        if abs(np.linalg.det(m4[:3, :3])) == 0.0 or not np.isfinite(m4).all():
            raise RuntimeError("rectangle: 'to_world' is singular or not finite")
        tw = ScalarTransform4f(np.concatenate([m4.ravel(), np.linalg.inv(m4).T.ravel()]).astype(np.float32))
        mesh = self.meshes[i]; rect = mesh["rect"]
        V = np.empty((4, 8), np.float32); F = np.empty((2, 4), np.uint32); n = np.empty(3, np.float32); ia = C.c_float()
        check(lib().har_shape_rectangle(_fp(tw.data), 1 if rect.get("flip") else 0, _fp(V), _up(F), _fp(n), C.byref(ia)))
        self._drop_handle()
        mesh["V"], mesh["F"] = V, F
        rect.update(to_world=tw, normal=n, inv_area=ia.value)
        em = mesh["emitter"]
        if em >= 0:
            e = self.emitters[em]
            e["to_world"] = tw.col_major_3x4(); e["normal"] = n; e["inv_area"] = ia.value
        self._drop_handle()

    @_static_table
    def _instance_keys(self):
        """'<instance>.to_world' (4 x 4, Instance::traverse, instance.cpp:79-85)"""
        return {k + ".to_world": i for i, k in enumerate(self.instance_keys)}

    def _set_instance_matrices_device(self, first, to_world):
        """params['<instance>.to_world'] as CUDA tensors + params.update(): `to_world` = device tensor (count, 12), column-major 3 x 4 rows for instances first .. first + count - 1.
        Inverses, shading / TLAS leaf records, instance bounds and the refit of the instance level are kernels on the current stream (har_scene_update_instances_device); the
        Python mirror self.instances is refreshed lazily (sync_host)."""
        torch = _torch()
        t = to_world.detach().to(torch.float32).contiguous()
        self._stale_instances = True
        self.device_instance_updates = getattr(self, "device_instance_updates", 0) + 1
        rc = lib().har_scene_update_instances_device(self._h, int(first), int(t.shape[0]), _ptr(t), _stream())
        if rc != 0:
            msg = (lib().har_last_error() or b"").decode()
            self._drop_handle(keep_geometry=False)
            raise RuntimeError(msg or "instance update failed")

    def _instance_matrix(self, i):
        self._sync_host_geometry()
        m = np.eye(4, dtype=np.float32); m[:3, :] = np.asarray(self.instances[i][1], np.float32).reshape(4, 3).T
        return m

    def _set_instance_matrix(self, i, m4):
        self._set_instance_matrices([(i, m4)])

    def _set_instance_matrices(self, items):
        """params['<instance>.to_world'] = ...; params.update(): the instance level of the acceleration structure is rebuilt IN PLACE (har_scene_update_instances:
        the bottom-level BVHs, the scene handle and every workspace stay; Scene::parameters_changed, scene.cpp:517-540 / scene_optix.inl:351-372) -- one call per
        run of consecutive instances"""
        self._sync_host_geometry()          # other instances may have moved on the device since: the mirror is read and rewritten below
        recs = {}
        for i, m4 in items:
            m = np.asarray(m4, np.float64).reshape(4, 4); inv = np.linalg.inv(m)
            tw = _f32(m[:3, :].T.reshape(-1)); to = _f32(inv[:3, :].T.reshape(-1))
            self.instances[i] = (self.instances[i][0], [float(x) for x in tw], [float(x) for x in to])
            recs[int(i)] = (tw, to)
        if self._h is None or not recs:
            return
        idx = sorted(recs); start = 0
        while start < len(idx):
            end = start
            while end + 1 < len(idx) and idx[end + 1] == idx[end] + 1:
                end += 1
            tw = np.ascontiguousarray(np.concatenate([recs[i][0] for i in idx[start:end + 1]]), np.float32)
            to = np.ascontiguousarray(np.concatenate([recs[i][1] for i in idx[start:end + 1]]), np.float32)
            rc = lib().har_scene_update_instances(self._h, idx[start], end - start + 1, _fp(tw), _fp(to), _stream())
            if rc != 0:          # the host mirror of the instance level may already differ from the device's: rebuild from self.instances at the next render
                msg = (lib().har_last_error() or b"").decode()
                self._drop_handle(keep_geometry=False)
                raise RuntimeError(msg or "instance update failed")
            start = end + 1

    def _set_vertex_positions(self, mesh, positions):
        """params['<shape>.positions'] = ... (a host tensor / array) + params.update(): normals regenerated on the host, the BLAS refitted on the device
        (har_scene_update_vertices); CUDA tensors take _set_vertex_positions_device instead"""
        V = self.meshes[mesh]["V"]
        V[:, :3] = np.asarray(positions, np.float32).reshape(V.shape[0], 3)
        if self.meshes[mesh]["flags"] & 1:
            # Mesh::parameters_changed (mesh.cpp:876-878): pack(regenerate_normals = positions written and normals not) -> compute_normals (:1216-1267)
            F = np.ascontiguousarray(self.meshes[mesh]["F"])
            if not V.flags["C_CONTIGUOUS"]:
                V = self.meshes[mesh]["V"] = np.ascontiguousarray(V)
            check(lib().har_mesh_compute_normals(V.shape[0], _fp(V), F.shape[0], _up(F)))
        if self._h is not None:
            # the BLAS that holds the mesh is REFITTED on the device (har_scene_update_vertices); a new scene only when the library asks for one: the mesh carries
            # an emitter (its sampling records are lowered from the positions), or the refitted tree has degraded (HAR_UPDATE_REBUILD_ADVISED: still valid, but a
            # fresh build traces faster -- the next render creates it)
            Vc = np.ascontiguousarray(self.meshes[mesh]["V"], np.float32)
            getattr(self, "_stale_meshes", set()).discard(mesh)
            self._after_vertex_update(mesh, lib().har_scene_update_vertices(self._h, int(mesh), _fp(Vc), _stream()))

    def _after_vertex_update(self, mesh, rc):
        """return codes of har_scene_update_vertices(_device): 2 / 3 = valid scene, but the next render needs / is better off with a new handle; anything else but 0 is an
        error AFTER the call may have touched the host mirror and the device arrays (allocation failure of the refit scratch, a HIP error mid-refit, a TLAS that no
        longer fits the traversal stacks): the handle is dropped so that the next render rebuilds from self.meshes instead of tracing a half-updated BVH"""
        if rc in (2, 3):
            if os.environ.get("HAR_VERBOSE"):
                import sys
                sys.stderr.write("[mitsuba3_amd] vertex update of mesh %d: %s -> new scene at the next render\n" % (mesh, (lib().har_last_error() or b"").decode() if rc == 2 else "refit cost ratio %.2f: rebuild advised" % self.refit_info()["ratio"]))
            self.accel_rebuilds = getattr(self, "accel_rebuilds", 0) + 1
            self._drop_handle()
        elif rc != 0:
            msg = (lib().har_last_error() or b"").decode()
            self._drop_handle(keep_geometry=False)
            raise RuntimeError(msg or "vertex update failed")

    def _set_vertex_positions_device(self, mesh, positions):
        """params['<shape>.positions'] (a CUDA tensor) + params.update(): the positions stay on the GPU -- vertex records, regenerated normals, shading triangles and the
        BLAS refit are kernels on the current stream (har_scene_update_vertices_device; Mesh::parameters_changed, mesh.cpp:848-899).  The numpy mirror meshes[mesh]['V'] is
        refreshed lazily (sync_host)."""
        torch = _torch()
        V = self.meshes[mesh]["V"]
        p = positions.detach().to(torch.float32).contiguous()
        if p.numel() != 3 * V.shape[0]:
            raise RuntimeError("positions: expected %d rows of 3 values" % V.shape[0])
        if not hasattr(self, "_stale_meshes"):
            self._stale_meshes = set()
        self._stale_meshes.add(mesh)
        self.device_vertex_updates = getattr(self, "device_vertex_updates", 0) + 1
        self._keep_positions = p                  # alive until the kernels that read it were enqueued on this stream (they were: the call below enqueues them)
        self._after_vertex_update(mesh, lib().har_scene_update_vertices_device(self._h, int(mesh), _ptr(p), _stream()))

    def refit_info(self):
        """(refits since the scene handle was created, cost figure of the last refit, its ratio to the first refit's, nodes)"""
        info = (C.c_double * 4)()
        check(lib().har_scene_refit_info(self._handle(), info))
        return dict(refits=int(info[0]), cost=info[1], ratio=info[2], nodes=int(info[3]), rebuilds=getattr(self, "accel_rebuilds", 0))

    @_static_table
    def _pose_keys(self):
        """the NON-differentiable placement parameters the reference's traverse() exposes (ParamFlags::NonDifferentiable): '<sensor>.to_world' (perspective.cpp:177,
        orthographic.cpp:95), '<emitter>.position' of a point light (point.cpp:86), '<emitter>.to_world' of spot and directional lights (spot.cpp:117,
        directional.cpp:96) -- 4 x 4 matrices / a 3-vector; params.update() re-lowers the sensor or the scene"""
        keys = {}
        for k, s in zip(self.sensor_keys, self.m_sensors):
            keys[k + ".to_world"] = ("sensor", s)
            if s.kind != 'orthographic':          # PerspectiveCamera::traverse (perspective.cpp:155-160)
                keys[k + ".x_fov"] = ("x_fov", s)
                keys[k + ".principal_point_offset_x"] = ("ppo_x", s); keys[k + ".principal_point_offset_y"] = ("ppo_y", s)
        for i, key in enumerate(self._emitter_order):
            t = self.emitters[i].get("type", 0)
            if t == 4:
                keys[key + ".position"] = ("position", i)
            elif t in (5, 6):
                keys[key + ".to_world"] = ("emitter_to_world", i)
            elif t == 2:        # EnvironmentMapEmitter::traverse (envmap.cpp:204-208): `scale`, `to_world` -- updatable here (the record is rebuilt with the next scene handle);
                                # and so is `data` (differentiable there, without a gradient here)
                keys[key + ".scale"] = ("env_scale", i); keys[key + ".to_world"] = ("emitter_to_world", i)
                keys[key + ".data"] = ("env_data", i)          # the texel tensor in the reference's layout: H x (W + 2) x 3, real column x at x + 1 (envmap.cpp:146-188)
            if t == 5:          # SpotLight::traverse (spot.cpp:115-116): the cone, in degrees -- updatable here; their gradient (the reference marks them Differentiable) is refused
                keys[key + ".cutoff_angle"] = ("cutoff_angle", i); keys[key + ".beam_width"] = ("beam_width", i)
            # Emitter::traverse (src/render/emitter.cpp:13): `sampling_weight`, NonDifferentiable; an area light is a child of its shape ('<shape>.emitter.*')
            keys[key + (".emitter" if t in (0, 3, 7) else "") + ".sampling_weight"] = ("sampling_weight", i)
            if t == 7:          # AreaLight::traverse -> "radiance" -> BitmapTexture::traverse: `data` and `to_uv` (bitmap.cpp:463-475).  `data` is Differentiable: the `prb`
                                # integrator's `light_texel_gradients` (switched on by mi.render for a key with requires_grad) produces its gradient
                keys[key + ".emitter.radiance.data"] = ("emitter_tex", i); keys[key + ".emitter.radiance.to_uv"] = ("emitter_to_uv", i)
        for b in self.bsdf_objs:            # BitmapTexture::traverse: `to_uv` (src/textures/bitmap.cpp), NonDifferentiable
            if b.texture is not None:
                keys[self._bsdf_key_base(b) + "." + b.slot0_name + ".to_uv"] = ("to_uv", b)
        return keys

    def _pose_value(self, kind, b):
        if kind == "sensor":
            return np.asarray(b.to_world.matrix, np.float32).reshape(4, 4).copy()
        if kind == "x_fov":
            return np.asarray([b.x_fov()], np.float32)
        if kind in ("ppo_x", "ppo_y"):
            return np.asarray([float(b.props.get('principal_point_offset_' + kind[-1], 0.0))], np.float32)
        if kind == "position":
            return np.asarray(self.emitters[b]["to_world"][9:12], np.float32).copy()
        if kind in ("cutoff_angle", "beam_width"):
            return np.asarray([self.emitters[b]["normal"][0 if kind == "cutoff_angle" else 1]], np.float32)
        if kind == "sampling_weight":
            return np.asarray([self.emitters[b].get("sampling_weight", 1.0)], np.float32)
        if kind == "env_scale":
            return np.asarray([self.emitters[b]["radiance"][0]], np.float32)
        if kind == "env_data":             # one halo column on each side carrying a copy of the opposite edge (envmap.cpp:146-188, refresh_halo)
            t = np.asarray(self.textures[self.emitters[b]["mesh"]], np.float32)
            return np.ascontiguousarray(np.concatenate([t[:, -1:], t, t[:, :1]], axis=1))
        if kind == "emitter_tex":
            return np.array(self.emitters[b]["light"].texture, np.float32)
        if kind in ("to_uv", "emitter_to_uv"):
            if kind == "emitter_to_uv":
                b = self.emitters[b]["light"]
            m = np.eye(3, dtype=np.float32)
            if b.tex_to_uv is not None:
                m[:2, :] = np.asarray(b.tex_to_uv, np.float32).reshape(2, 3)
            return m
        m = np.eye(4, dtype=np.float32); m[:3, :] = np.asarray(self.emitters[b]["to_world"], np.float32).reshape(4, 3).T
        return m

    def _validate_spots(self):
        for e in self.emitters:                                      # SpotLight::update, spot.cpp:300-306
            if e.get("type") == 5 and not (e["normal"][0] >= e["normal"][1] and e["normal"][0] > 0):
                raise RuntimeError("spot: cutoff_angle must be positive and not smaller than beam_width")
        pending = self.__dict__.get("_emitters_to_push")
        if pending:                                                  # spot cones whose pair of angles is complete and valid: the record goes to the scene in place
            todo = sorted(pending); pending.clear()
            if self._h is not None:
                for b in todo:
                    self._push_delta_emitter(b)
        if getattr(self, "_weights_dirty", False):                   # Scene::parameters_changed -> update_emitter_sampling_distribution (scene.cpp:523-528)
            self._weights_dirty = False
            if self._h is not None:
                w = _f32([e.get("sampling_weight", 1.0) for e in self.emitters])
                check(lib().har_scene_set_emitter_sampling_weights(self._h, _fp(w), len(self.emitters)))

    def sample_emitter(self, index_sample, active=True):
        """Scene::sample_emitter(index_sample, active) -> (index, emitter_weight, reused sample) (src/render/scene.cpp:248-271), array-valued"""
        torch = _torch(); dev = _device()
        s = torch.as_tensor(index_sample, dtype=torch.float32, device=dev).reshape(-1).contiguous(); n = s.numel()
        index = torch.empty(n, dtype=torch.int32, device=dev); weight = torch.empty(n, dtype=torch.float32, device=dev); reused = torch.empty_like(weight)
        mask = _mask(active, n)
        check(lib().har_scene_sample_emitter(self._handle(), n, _ptr(s), _ptr(mask), _ptr(index), _ptr(weight), _ptr(reused), _stream()))
        return index, weight, reused

    def pdf_emitter(self, index, active=True):
        """Scene::pdf_emitter(index, active) (scene.cpp:273-279)"""
        torch = _torch(); dev = _device()
        i = torch.as_tensor(index, device=dev).to(torch.int32).reshape(-1).contiguous(); n = i.numel()
        pdf = torch.empty(n, dtype=torch.float32, device=dev)
        mask = _mask(active, n)
        check(lib().har_scene_pdf_emitter(self._handle(), n, _ptr(i), _ptr(mask), _ptr(pdf), _stream()))
        return pdf

    def _set_pose(self, kind, b, value):
        # validate first, assign last: a rejected value leaves the scene (and the params entry's counterpart) as it was
        if kind == "sensor":
            m = np.asarray(value, np.float64).reshape(4, 4)
            if not np.isfinite(m).all() or abs(np.linalg.det(m)) < 1e-30:
                raise RuntimeError("sensor to_world: the matrix is singular or not finite")
            new = ScalarTransform4f(np.concatenate([m.astype(np.float32).ravel(), np.linalg.inv(m).T.astype(np.float32).ravel()]))
            if b.kind != 'orthographic' and new.has_scale():
                raise RuntimeError("Scale factors in the camera-to-world transformation are not allowed!")
            old = b.to_world
            b.to_world = new
            try:
                b.update()                                       # the sensor record travels with every render call: no scene handle involved
            except Exception:
                b.to_world = old; b.update()
                raise
            return
        if kind in ("ppo_x", "ppo_y"):       # perspective.cpp:158-159: the principal point, a fraction of the film size; the projection is re-lowered
            v = float(np.asarray(value, np.float32).reshape(-1)[0])
            if not math.isfinite(v):
                raise RuntimeError("perspective: the principal point offset is not finite")
            b.props['principal_point_offset_' + kind[-1]] = v
            b.update()
            return
        if kind == "x_fov":                  # PerspectiveCamera::parameters_changed -> update_camera_transforms (perspective.cpp:163-198): the projection follows the new angle
            fov = float(np.asarray(value, np.float32).reshape(-1)[0])
            if not (0.0 < fov < 180.0):
                raise RuntimeError("The horizontal field of view must be in the range [0, 180]!")
            old = dict(b.props)
            b.props.pop('focal_length', None); b.props['fov'] = fov; b.props['fov_axis'] = 'x'
            try:
                b.update()
            except Exception:
                b.props = old; b.update()
                raise
            return
        if kind == "sampling_weight":
            w = float(np.asarray(value, np.float32).reshape(-1)[0])
            if not (w >= 0.0) or not math.isfinite(w):
                raise RuntimeError("DiscreteDistribution: entries must be non-negative!")
            e = dict(self.emitters[b]); e["sampling_weight"] = w; self.emitters[b] = e
            self._weights_dirty = True               # the distribution is rebuilt once, after every weight of this update() is in (_validate_spots)
            return
        if kind == "emitter_tex":      # BitmapTexture::parameters_changed -> rebuild_internals (bitmap.cpp:484-493): new texels, new texel distribution
            light = self.emitters[b]["light"]
            v = np.ascontiguousarray(np.asarray(value, np.float32))
            if v.shape != light.texture.shape or not np.isfinite(v).all():
                raise RuntimeError("area: the radiance bitmap must keep its shape %s and be finite" % (light.texture.shape,))
            if not (v >= 0).all() or float(v.sum()) <= 0.0:
                raise RuntimeError("area: the radiance bitmap must be non-negative with some luminance to sample")
            light.texture = v; self.textures[light.tex_index] = v
            if self._h is not None:
                check(lib().har_scene_set_texture(self._h, light.tex_index, _fp(v)))
            return
        if kind in ("to_uv", "emitter_to_uv"):
            if kind == "emitter_to_uv":
                b = self.emitters[b]["light"]
            m = np.asarray(value.matrix if isinstance(value, (ScalarTransform3f, ScalarTransform4f)) else value, np.float32)
            if m.shape == (4, 4):
                m = np.array([[m[0, 0], m[0, 1], m[0, 3]], [m[1, 0], m[1, 1], m[1, 3]], [0.0, 0.0, 1.0]], np.float32)
            m = m.reshape(3, 3)
            if not np.isfinite(m).all() or float(np.linalg.det(m[:2, :2].astype(np.float64))) == 0.0:
                raise RuntimeError("bitmap: 'to_uv' is singular")
            rows = [float(x) for x in m[:2, :].reshape(-1)]
            if kind == "emitter_to_uv":
                _check_sampling_transform(rows)
            b.tex_to_uv = rows; self.texture_to_uv[b.tex_index] = rows
            if self._h is not None:
                check(lib().har_scene_set_texture_to_uv(self._h, b.tex_index, _fp(_f32(rows))))
            return
        if kind == "env_data":             # EnvironmentMapEmitter::parameters_changed (envmap.cpp:208-256): the real columns are what was written, the halo is refreshed from
            v = np.asarray(value, np.float32)  # them, the sampling distribution is rebuilt (here: with the next scene handle, from the texture table)
            if v.ndim != 3:
                raise RuntimeError("Environment map data has dimension %d, expected 3" % v.ndim)
            if v.shape[2] != 3:
                raise RuntimeError("Environment map data has %d channels, expected 3" % v.shape[2])
            if v.shape[1] < 4 or v.shape[0] < 3 or not np.isfinite(v).all():
                raise RuntimeError("Environment map data: at least 3 rows and 2 real columns of finite values")
            self._drop_handle()
            self.textures[self.emitters[b]["mesh"]] = np.ascontiguousarray(v[:, 1:-1, :])
            self._drop_handle()
            return
        e = dict(self.emitters[b])
        if kind == "env_scale":
            sc = float(np.asarray(value, np.float32).reshape(-1)[0])
            if not math.isfinite(sc):
                raise RuntimeError("envmap: 'scale' is not finite")
            e["radiance"] = [sc] + [float(x) for x in e["radiance"][1:]]
        elif kind in ("cutoff_angle", "beam_width"):
            nrm = [float(x) for x in e["normal"]]; nrm[0 if kind == "cutoff_angle" else 1] = float(np.asarray(value, np.float32).reshape(-1)[0])
            e["normal"] = nrm                                        # the pair is validated once both values of an update() are in (_validate_spots)
        elif kind == "position":
            pos = np.asarray(value, np.float32).reshape(3)
            if not np.isfinite(pos).all():
                raise RuntimeError("emitter position is not finite")
            e["to_world"] = list(e["to_world"][:9]) + [float(x) for x in pos]
        else:
            m = np.asarray(value, np.float64).reshape(4, 4)
            if not np.isfinite(m).all() or abs(np.linalg.det(m)) < 1e-30:
                raise RuntimeError("emitter to_world: the matrix is singular or not finite")
            inv = np.linalg.inv(m)
            e["to_world"] = [float(x) for x in m[:3, :].T.reshape(-1)]; e["to_local"] = [float(x) for x in inv[:3, :].T.reshape(-1)]
        self.emitters[b] = e
        if self._h is not None and e.get("type", 0) in (4, 5, 6) and kind not in ("cutoff_angle", "beam_width"):
            # the record of a delta emitter is re-lowered IN PLACE (har_scene_set_delta_emitter): the scene handle and the acceleration data survive a moved light.
            # (cone values arrive one by one and are validated as a pair at the end of update(): _validate_spots pushes the record then)
            self._push_delta_emitter(b)
        elif self._h is not None and e.get("type", 0) in (4, 5, 6):
            self.__dict__.setdefault("_emitters_to_push", set()).add(b)
        else:
            self._drop_handle()                                  # other emitter records are part of the scene's tables: rebuilt with the next handle

    def _push_delta_emitter(self, b):
        rec = _capi.HarEmitter()
        self._fill_emitter_record(rec, self.emitters[b])
        rc = lib().har_scene_set_delta_emitter(self._h, int(b), C.byref(rec))
        if rc != 0:
            msg = (lib().har_last_error() or b"").decode()
            self._drop_handle()
            raise RuntimeError(msg or "emitter update failed")

    def _gradients(self, g_refl, g_tex, g_emit=None):
        out = {}
        for k, (kind, b) in self._param_keys().items():
            if kind == "emit":
                if g_emit is not None:
                    out[k] = g_emit[b]
            else:
                out[k] = g_tex[b.tex_index] if kind == "tex" else g_refl[b.index]
        return out


class ParamFlags:
    """include/mitsuba/core/object.h:363-372"""
    Differentiable = 0
    NonDifferentiable = 1
    Discontinuous = 2
    ReadOnly = 4


class SceneParameters(dict):
    """mi.traverse(scene): differentiable parameters as torch tensors (util.py SceneParameters)."""

    def flags(self, key):
        """SceneParameters.flags (util.py:146-148): what THIS variant can do with the entry -- Differentiable where `prb` produces its gradient (| Discontinuous where the
        reference's traverse() says so: geometry, roughness, complex IOR, texels of a light), NonDifferentiable for placements and plain values, ReadOnly for what is only shown"""
        if key not in self:
            raise KeyError(key)
        sc = self.scene
        if key in self._read_only:
            return ParamFlags.ReadOnly | ParamFlags.NonDifferentiable
        if key in sc._param_keys():
            return ParamFlags.Differentiable
        if key in sc._bsdf_param_keys():
            return ParamFlags.NonDifferentiable if sc._bsdf_param_keys()[key][0] == "ior" else (ParamFlags.Differentiable if sc._bsdf_param_keys()[key][0] == "slot1" else ParamFlags.Differentiable | ParamFlags.Discontinuous)
        if key in sc._position_keys() or key in sc._instance_keys() or key in sc._rect_keys():
            return ParamFlags.Differentiable | ParamFlags.Discontinuous
        if sc._pose_keys().get(key, (None,))[0] == "emitter_tex":
            return ParamFlags.Differentiable | ParamFlags.Discontinuous
        return ParamFlags.NonDifferentiable

    def set_dirty(self, key):
        """SceneParameters.set_dirty (util.py:150-185): the next update() treats the entry as written (for writes the version counter does not see)"""
        if key not in self:
            raise KeyError(key)
        self._written.add(key)

    def keep(self, keys):
        """SceneParameters.keep (util.py:238-256): only the entries whose names match one of `keys` (regular expressions, `re.match`) stay in the table"""
        import re
        if not isinstance(keys, list):
            keys = [keys]
        regexps = [re.compile(k).match for k in keys]
        for k in [k for k in list(self.keys()) if not any(r(k) for r in regexps)]:
            dict.__delitem__(self, k)
            self._read_only.discard(k); self._written.discard(k)
        self._host_table = None            # the update tables are rebuilt over the entries that are left

    def __init__(self, scene):
        super().__init__()
        torch = _torch()
        self.scene = scene
        scene.sync_host()
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        for k, (kind, b) in scene._param_keys().items():
            value = scene.emitters[b]["radiance"] if kind == "emit" else (b.texture if kind == "tex" else b.value)
            self[k] = torch.tensor(np.asarray(value, np.float32), dtype=torch.float32, device=dev)
        for k, (what, b) in scene._bsdf_param_keys().items():
            self[k] = torch.tensor(np.asarray(scene._bsdf_param_value(what, b), np.float32), dtype=torch.float32, device=dev)
        for k, m in scene._position_keys().items():
            self[k] = torch.tensor(np.ascontiguousarray(scene.meshes[m]["V"][:, :3]), dtype=torch.float32, device=dev)            # N x 3, as the reference's `positions` tensor
        for k, i in scene._instance_keys().items():
            self[k] = torch.tensor(scene._instance_matrix(i), dtype=torch.float32, device=dev)
        for k, i in scene._rect_keys().items():
            self[k] = torch.tensor(scene._rect_matrix(i), dtype=torch.float32, device=dev)
        for k, (kind, b) in scene._pose_keys().items():
            self[k] = torch.tensor(scene._pose_value(kind, b), dtype=torch.float32, device=dev)
        # what Mesh::traverse (src/render/mesh.cpp:822-843) also registers and this variant can only SHOW: the index buffer and the texture coordinates (N x 2) of the
        # triangle meshes -- readable under the reference's names, refused on write like a ParamFlags::ReadOnly entry (util.py:59-60)
        self._read_only = set()
        for k, m in scene._position_keys().items():
            mesh = scene.meshes[m]; base = k[:-len(".positions")]
            self[base + ".faces"] = torch.tensor(np.ascontiguousarray(mesh["F"][:, :3]).astype(np.int64), dtype=torch.int64, device=dev)
            self._read_only.add(base + ".faces")
            if mesh["flags"] & 2:
                self[base + ".texcoords"] = torch.tensor(np.ascontiguousarray(mesh["V"][:, 6:8]), dtype=torch.float32, device=dev)
                self._read_only.add(base + ".texcoords")
        # ... and of the sensors: what ProjectiveCamera / Sensor / Film register next to `to_world` (sensor.h:135-141,206-210, film.cpp:55-57), all NonDifferentiable there;
        # shown, not updatable here (a new film size is a new scene description)
        for k, sn in zip(scene.sensor_keys, scene.m_sensors):
            f = sn.film()
            for name, val in (("near_clip", [sn.near_clip]), ("far_clip", [sn.far_clip]), ("shutter_open", [0.0]), ("shutter_open_time", [0.0]),
                              ("film.size", [f.width, f.height]), ("film.crop_size", list(f.crop_size_)), ("film.crop_offset", list(f.crop_offset_))):
                key = k + "." + name
                self[key] = torch.tensor(val, dtype=torch.float32 if "clip" in name or "shutter" in name else torch.int64, device=dev)
                self._read_only.add(key)
        self._written = set()           # keys assigned since the last update() (SceneParameters.__setitem__ flags them in the reference, util.py)
        # the tensors above ARE the scene's values: recorded as applied, so that the first update() touches only what was written or stepped since (vertex positions on
        # the GPU are never compared -- a mesh is updated when its tensor's version counter moved, see _changed_keys)
        snap = self.__dict__.setdefault("_snapshot", {}); seen = self.__dict__.setdefault("_seen", {})
        for k, what, _ in self._host_kinds():
            t = self[k]
            seen[k] = (t, t._version)
            snap[k] = None if (what in ("pos", "inst") and t.is_cuda) else t.detach().clone()

    def __setitem__(self, key, value):
        if hasattr(self, "_written"):                 # (the constructor fills the table before `_written` exists)
            if key not in self:
                raise KeyError(key)                   # util.py:57: `self.properties[key]` -- a name traverse() did not register is not a parameter
            if key in self._read_only:
                raise Exception("%s is a read-only parameter!" % key)      # util.py:59-60
        super().__setitem__(key, value)
        if hasattr(self, "_written"):
            self._written.add(key)

    def _host_kinds(self):
        """(key, setter) of the parameters whose update runs on the host (geometry: accel update; poses; the non-colour BSDF parameters, whose records are
        re-lowered): the key tables are those of the scene at traverse() time -- the set of parameters of a scene does not change"""
        if getattr(self, "_host_table", None) is None:
            sc = self.scene; t = []
            for k, m in sc._position_keys().items():
                t.append((k, "pos", m))
            for k, i in sc._instance_keys().items():
                t.append((k, "inst", i))
            for k, i in sc._rect_keys().items():
                t.append((k, "rect", i))
            for k, (kind, b) in sc._pose_keys().items():
                t.append((k, "pose", (kind, b)))
            for k, (what, b) in sc._bsdf_param_keys().items():
                t.append((k, "bsdf", (what, b)))
            self._host_table = [e for e in t if e[0] in self]                      # (keep() may have dropped entries)
            self._colour_table = [e for e in sc._param_keys().items() if e[0] in self]
        return self._host_table

    def _changed_keys(self, written):
        """which host-updated keys hold other values than at the last update(): WRITTEN keys (SceneParameters.__setitem__ flags them, util.py) and tensors modified
        in place (an optimiser step).  A tensor that is the same object with the same version counter as at the last update() is unchanged without looking at it;
        the others are compared with the snapshot of the last update() ON THE TENSOR'S DEVICE, one flag per key, ONE read-back for all of them (round 4 copied
        every parameter to the host, every step).  Vertex positions that live on the GPU are not even compared: a new version counter IS the change (the device
        update costs less than the comparison's synchronisation).  Values without a version counter (numpy arrays, lists) are compared every call.
        (Writes that bypass the version counter -- `tensor.data.add_()`, raw pointers -- need `params[key] = params[key]`, as every update does in the reference.)
        Returns (changed keys, marks): `marks[k]()` records key k as applied -- update() calls it AFTER the key's setter succeeded, so that a rejected value
        (singular matrix, wrong size, a C error) is looked at again by the next update() instead of being remembered as pushed."""
        torch = _torch()
        table = self._host_kinds()
        snap = self.__dict__.setdefault("_snapshot", {})
        seen = self.__dict__.setdefault("_seen", {})          # key -> (the tensor object, its version counter) at the last update()
        changed = set(k for k, _, _ in table if k in written or k not in snap)
        flags = []; names = []
        device_pos = set()
        for k, what, _ in table:
            t = self[k]
            if what in ("pos", "inst") and getattr(t, "is_cuda", False) and self.scene._h is not None and not os.environ.get("HAR_HOST_VERTEX_UPDATE"):
                device_pos.add(k)
                mark = seen.get(k)
                if mark is None or mark[0] is not t or mark[1] != t._version:
                    changed.add(k)
                continue
            if k in changed:
                continue
            if not hasattr(t, "detach") or not hasattr(t, "_version"):
                changed.add(k); continue
            mark = seen.get(k)
            if mark is not None and mark[0] is t and mark[1] == t._version:
                continue                    # the same tensor object, never written in place since (torch bumps _version on every in-place op): nothing to compare
            t = t.detach(); old = snap[k]
            if old is None or old.shape != t.shape or old.device != t.device or old.dtype != t.dtype:
                changed.add(k); continue
            flags.append((t != old).any()); names.append(k)
        if flags:
            by_dev = {}
            for f, k in zip(flags, names):
                by_dev.setdefault(f.device, []).append((f, k))
            for dev, items in by_dev.items():
                got = torch.stack([f for f, _ in items]).tolist()          # one synchronisation per device
                changed.update(k for (_, k), g in zip(items, got) if g)

        def marker(k):
            t = self[k]

            def apply():
                mark = seen.get(k)
                if mark is not None and mark[0] is t and hasattr(t, "_version") and mark[1] == t._version and k in snap:
                    return                       # the same tensor at the same version as at the last update(): the snapshot and the mark are current (no clone per key and step)
                if k in device_pos:
                    snap[k] = None               # never compared (see above): no clone of a million vertices per step, no read-back of a flag
                else:
                    snap[k] = t.detach().clone() if hasattr(t, "detach") else torch.as_tensor(np.array(t, np.float32, copy=True))
                if hasattr(t, "_version"):
                    seen[k] = (t, t._version)
            return apply
        marks = {k: marker(k) for k, _, _ in table}
        return changed, marks

    def update(self, values=None):
        """SceneParameters.update (src/python/python/util.py): push the values to the scene.  Colours, bitmaps, emitter radiances AND VERTEX POSITIONS that live on the
        GPU go to the scene's device arrays directly (har_scene_set_*_device, har_scene_update_vertices_device: device-to-device on the current stream, no host round
        trip, no synchronisation); instance transforms, poses and the non-colour BSDF parameters are compared on their device and only the ones that changed travel to
        the host (accel update / re-lowering)."""
        if values:
            for k, v in values.items():
                if k in self:                  # util.py:210-213: names that are not parameters are skipped
                    self[k] = v
        torch = _torch()
        written, self._written = self._written, set()
        sc = self.scene
        changed, marks = self._changed_keys(written)
        moved = []; moved_keys = []; moved_dev = []
        for k, what, ref in self._host_kinds():
            if k not in changed:
                if what not in ("pos", "inst") or not getattr(self[k], "is_cuda", False):
                    marks[k]()              # unchanged: remember the (possibly new) tensor object and version
                continue
            v = self[k]
            if what == "inst" and getattr(v, "is_cuda", False) and sc._h is not None and not os.environ.get("HAR_HOST_VERTEX_UPDATE"):
                moved_dev.append((ref, k, v))
                continue
            if what == "pos" and getattr(v, "is_cuda", False) and sc._h is not None and not os.environ.get("HAR_HOST_VERTEX_UPDATE"):
                sc._set_vertex_positions_device(ref, v)
                marks[k]()
                continue
            v = v.detach().to("cpu").numpy().astype(np.float32) if hasattr(v, "detach") else np.asarray(v, np.float32)
            if what == "pos":
                v = v.reshape(-1, 3)
                sc._sync_host_geometry()
                # a WRITTEN key notifies the mesh even when the values are the old ones: the vertex normals are regenerated (mesh.cpp:876-878)
                if k in written or not np.array_equal(v, sc.meshes[ref]["V"][:, :3]):
                    sc._set_vertex_positions(ref, v)
            elif what == "inst":
                if not np.array_equal(v.reshape(4, 4), sc._instance_matrix(ref)):
                    moved.append((ref, v.reshape(4, 4))); moved_keys.append(k)
                    continue                # marked once the whole run of instances went through
            elif what == "rect":
                if not np.array_equal(v.reshape(4, 4), sc._rect_matrix(ref)):
                    sc._set_rect_to_world(ref, v.reshape(4, 4))
            elif what == "pose":
                if not np.array_equal(v.reshape(-1), sc._pose_value(*ref).reshape(-1)):
                    sc._set_pose(ref[0], ref[1], v)
            else:
                if not np.array_equal(v.reshape(-1), np.asarray(sc._bsdf_param_value(*ref), np.float32).reshape(-1)):
                    sc._set_bsdf_param(ref[0], ref[1], v.reshape(-1))
            marks[k]()
        if moved:
            sc._set_instance_matrices(moved)
            for k in moved_keys:
                marks[k]()
        if moved_dev and sc._h is None:          # a later key of this update() (a rectangle's to_world) asked for a new scene: the matrices go to the host mirrors it is built from
            sc._set_instance_matrices([(ref, v.detach().to("cpu").numpy().astype(np.float32).reshape(4, 4)) for ref, _, v in moved_dev])
            for _, k, _ in moved_dev:
                marks[k]()
            moved_dev = []
        if moved_dev:          # instance transforms that live on the GPU: one call per run of consecutive instances, nothing leaves the device
            moved_dev.sort(key=lambda e: e[0]); start = 0
            while start < len(moved_dev):
                end = start
                while end + 1 < len(moved_dev) and moved_dev[end + 1][0] == moved_dev[end][0] + 1:
                    end += 1
                rows = torch.stack([e[2].detach().to(torch.float32).reshape(4, 4)[:3, :].T.reshape(-1) for e in moved_dev[start:end + 1]])
                sc._set_instance_matrices_device(moved_dev[start][0], rows)
                for e in moved_dev[start:end + 1]:
                    marks[e[1]]()
                start = end + 1
        sc._validate_spots()
        stream = None
        pushed = self.__dict__.setdefault("_pushed", {})      # key -> (tensor object, version counter, scene handle) of the last push
        for k, (kind, b) in self._colour_table:
            t = self[k]
            mark = pushed.get(k)
            if k not in written and mark is not None and hasattr(t, "_version") and mark[0] is t and mark[1] == t._version and mark[2] == (sc._h.value if sc._h is not None else None):
                continue                    # the value the scene already holds (values without a version counter -- numpy arrays -- are pushed every call)
            new_mark = (t, getattr(t, "_version", None), sc._h.value if sc._h is not None else None)      # recorded once the push below went through
            on_gpu = hasattr(t, "is_cuda") and t.is_cuda and sc._h is not None
            if on_gpu:
                # the scene's copy is the value AT update(): snapshot on the device (a later in-place edit of the tensor is not an update), host mirror refreshed lazily
                v = t.detach().to(torch.float32).contiguous()
                stream = stream or _stream()
                if kind == "emit":
                    if v.numel() != 3:
                        raise RuntimeError("%s: expected 3 values" % k)
                    check(lib().har_scene_set_emitter_radiance_device(sc._h, b, _ptr(v), stream))
                elif kind == "tex":
                    if v.numel() != b.texture.size:
                        raise RuntimeError("%s: expected a tensor of shape %s" % (k, b.texture.shape))
                    check(lib().har_scene_set_texture_device(sc._h, b.tex_index, _ptr(v), stream))
                else:
                    if v.numel() != 3:
                        raise RuntimeError("%s: expected 3 values" % k)
                    check(lib().har_scene_set_reflectance_device(sc._h, b.index, _ptr(v), stream))
                sc._device_values[(kind, b if kind == "emit" else (b.tex_index if kind == "tex" else b.index))] = (v if v.data_ptr() != t.data_ptr() else v.clone(), b)
                pushed[k] = new_mark
                continue
            v = t.detach().to("cpu").numpy().astype(np.float32) if hasattr(t, "detach") else np.asarray(t, np.float32)
            if kind == "emit":
                sc._device_values.pop(("emit", b), None)
                sc.emitters[b]["radiance"] = np.ascontiguousarray(v.reshape(3))
                if sc._h is not None:
                    check(lib().har_scene_set_emitter_radiance(sc._h, b, _fp(sc.emitters[b]["radiance"])))
            elif kind == "tex":
                sc._device_values.pop(("tex", b.tex_index), None)
                b.texture = np.ascontiguousarray(v.reshape(b.texture.shape)); sc.textures[b.tex_index] = b.texture
                if sc._h is not None:
                    check(lib().har_scene_set_texture(sc._h, b.tex_index, _fp(b.texture)))
            else:
                sc._device_values.pop(("rgb", b.index), None)
                b.value = np.ascontiguousarray(v.reshape(3))
                if sc._h is not None:
                    check(lib().har_scene_set_reflectance(sc._h, b.index, _fp(b.value)))
            pushed[k] = new_mark


class ObjectParameters:
    """mi.traverse(<a plugin object>): the reference walks any Object (util.py:263-341); its tests traverse emitters, shapes and BSDFs on their own and read the entries without
    a prefix (`params['cutoff_angle']`, test_spot.py:189-212).  The object is placed in a private scene under the name `obj`; this view strips and adds that prefix."""

    def __init__(self, obj):
        if isinstance(obj, BSDF):
            if obj.scene is None:
                Scene({'obj': obj})
            self.scene = obj.scene
            self._prefix = (obj.id if obj.id else "bsdf%d" % obj.index) + "."
        else:
            # one private scene per object, kept with it: a second traverse() of the same object sees what the first one's update() did (the reference updates the object itself)
            sc = getattr(obj, '_traverse_scene', None)
            if sc is None:
                sc = Scene({'obj': obj})
                try:
                    obj._traverse_scene = sc
                except AttributeError:
                    pass
            self.scene = sc
            self._prefix = "obj."
        self._inner = SceneParameters(self.scene)
        self._inner.keep(["^" + re.escape(self._prefix)])

    def keys(self):
        return [k[len(self._prefix):] for k in self._inner.keys()]

    def items(self):
        return [(k[len(self._prefix):], v) for k, v in self._inner.items()]

    def __iter__(self):
        return iter(self.items())

    def __len__(self):
        return len(self._inner)

    def __contains__(self, key):
        return self._prefix + key in self._inner

    def __getitem__(self, key):
        return self._inner[self._prefix + key]

    def __setitem__(self, key, value):
        self._inner[self._prefix + key] = value

    def flags(self, key):
        return self._inner.flags(self._prefix + key)

    def set_dirty(self, key):
        self._inner.set_dirty(self._prefix + key)

    def keep(self, keys):
        keys = keys if isinstance(keys, list) else [keys]
        regexps = [re.compile(k).match for k in keys]
        self._inner.keep(["^" + re.escape(self._prefix + k) + "$" for k in self.keys() if any(r(k) for r in regexps)])

    def update(self, values=None):
        return self._inner.update({self._prefix + k: v for k, v in values.items()} if values else None)


def traverse(node):
    """mi.traverse (util.py:263-341): the parameter table of a scene, or of a single plugin object (names without a prefix)"""
    return SceneParameters(node) if isinstance(node, Scene) else ObjectParameters(node)


class DeviceGroup:
    """N GPUs driven by ONE host thread through the C ABI (har_multi_*: a replica of the scene, an integrator and a stream per device; row bands with global lane
    indices; one RCCL reduce of the film on devices[0]).  The single-call counterpart of render_distributed (one process per GPU, torch.distributed), for hosts that
    call Integrator::render once from one thread (include/mitsuba/render/integrator.h:74-79).  `path` renders; parameters are those of the scene at construction
    (replica(k) hands out the per-device handles for updates)."""

    def __init__(self, scene, devices=(0,), integrator=None):
        integrator = integrator or scene.integrator()
        if integrator is None:
            raise Exception('No integrator specified!')
        _device()
        self.scene, self.integrator, self.devices = scene, integrator, [int(d) for d in devices]
        d = scene.desc(); h = C.c_void_p()
        devs = (C.c_int * len(self.devices))(*self.devices)
        check(lib().har_multi_create(C.byref(d), 0 if integrator.type == 'path' else 1, integrator.max_depth, integrator.rr_depth, integrator.chunk_lanes or 0,
                                     devs, len(self.devices), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                lib().har_multi_destroy(self._h); self._h = None
        except Exception:
            pass

    def render(self, sensor=0, seed=0, spp=0, develop=True):
        torch = _torch()
        s = self.scene.sensors()[sensor] if isinstance(sensor, int) else sensor
        if spp:
            s.sampler().set_sample_count(spp)
        spp = s.sampler().sample_count()
        w, h = s.film().crop_size()
        dev = torch.device("cuda", self.devices[0])
        with torch.cuda.device(dev):
            colour = getattr(s.film(), "colour", 0)
            out = torch.empty((h, w, 1 if colour == 1 else 3) if develop else (h, w, 4), dtype=torch.float32, device=dev)
            check(lib().har_multi_render(self._h, C.byref(s.har), (s.sampler().m_base_seed + int(seed)) & 0xffffffff, spp, colour,
                                         _ptr(out) if develop else None, None if develop else _ptr(out), _stream()))
        return out

    def render_backward(self, grad_in, sensor=0, seed=0, spp=0):
        """RBIntegrator.render_backward over the group (har_multi_render_backward): {key: gradient tensor on devices[0]} for the colour, bitmap and -- with the
        integrator's `emitter_gradients` -- emitter-radiance parameters"""
        if self.integrator.type != 'prb':
            raise RuntimeError("render_backward(): only the `prb` integrator implements the adjoint pass in hip_ad_rgb")
        torch = _torch()
        s = self.scene.sensors()[sensor] if isinstance(sensor, int) else sensor
        if spp:
            s.sampler().set_sample_count(spp)
        spp = s.sampler().sample_count()
        w, h = s.film().crop_size()
        if getattr(s.film(), "colour", 0) or getattr(s.film(), "alpha", False):
            raise RuntimeError("DeviceGroup.render_backward(): rgb films only")
        sc = self.scene
        dev = torch.device("cuda", self.devices[0])
        with torch.cuda.device(dev):
            g_in = torch.as_tensor(grad_in, dtype=torch.float32, device=dev).reshape(h, w, 3).contiguous()
            g_refl = torch.zeros((len(sc.bsdfs), 3), dtype=torch.float32, device=dev)
            g_tex = [torch.zeros(tuple(t.shape), dtype=torch.float32, device=dev) for t in sc.textures]
            ptrs = (C.c_void_p * max(1, len(g_tex)))(*[t.data_ptr() for t in g_tex])
            g_emit = torch.zeros((max(1, len(sc.emitters)), 3), dtype=torch.float32, device=dev) if self.integrator.emitter_gradients else None
            check(lib().har_multi_render_backward(self._h, C.byref(s.har), _ptr(g_in), (s.sampler().m_base_seed + int(seed)) & 0xffffffff, spp, _ptr(g_refl), ptrs,
                                                  _ptr(g_emit) if g_emit is not None else None, _stream()))
        return sc._gradients(g_refl, g_tex, g_emit)

    def replica(self, k):
        sc = C.c_void_p(); it = C.c_void_p(); dev = C.c_int()
        check(lib().har_multi_replica(self._h, int(k), C.byref(sc), C.byref(it), C.byref(dev)))
        return sc, it, dev.value

    def stats(self):
        """device counters summed over the replicas (paths, vertices, rays)"""
        tot = {}
        for k in range(len(self.devices)):
            _, it, _ = self.replica(k)
            st = _capi.HarStats()
            check(lib().har_render_stats(it, C.byref(st)))
            for f, _t in st._fields_:
                tot[f] = tot.get(f, 0) + int(getattr(st, f))
        return tot

    def info(self):
        n = len(self.devices)
        rows = (C.c_uint32 * (n + 1))(); ms = (C.c_float * n)(); note = C.create_string_buffer(160); nd = C.c_uint32()
        check(lib().har_multi_info(self._h, C.byref(nd), rows, ms, note, 160))
        return dict(devices=list(self.devices), band_rows=list(rows), band_ms=list(ms), reduce=note.value.decode())


# ---------------------------------------------------------------------------
#  PluginManager (src/core/plugin.cpp:157-282): registry keyed by (name, variant)
# ---------------------------------------------------------------------------

_REGISTRY = {}
_BSDF_PLUGINS = ('diffuse', 'dielectric', 'conductor', 'plastic', 'roughconductor', 'roughplastic', 'twosided')


def register_plugin(name, variant_name, instantiate):
    """PluginManager::register_plugin (plugin.cpp:209-214)."""
    _REGISTRY[(name, variant_name)] = instantiate


def _resolve(value, named, parent_key=None):
    if isinstance(value, dict) and 'type' in value:
        if value['type'] == 'ref':
            if value['id'] not in named:
                raise RuntimeError("Referenced id \"%s\" not found" % value['id'])
            return named[value['id']]
        return _create(value, named, parent_key)
    return value


def _create(props, named, key=None):
    t = props['type']
    ctor = _REGISTRY.get((t, VARIANT))
    if ctor is None:
        raise RuntimeError("Plugin \"%s\" not found for variant \"%s\". Available: %s" % (t, VARIANT, sorted(k[0] for k in _REGISTRY)))
    return ctor(props, named, key)


def _shape_common(m, props, named):
    for k, v in props.items():
        if isinstance(v, dict) and v.get('type') in ('ref',) or isinstance(v, dict) and v.get('type') in _BSDF_PLUGINS:
            obj = _resolve(v, named, k)
            if isinstance(obj, BSDF):
                m.bsdf = obj
        elif isinstance(v, BSDF):
            m.bsdf = v
        elif isinstance(v, dict) and v.get('type') == 'area':
            if m.emitter is not None:
                raise RuntimeError("Only a single Emitter child object can be specified per shape.")       # shape.cpp:25-27
            a = AreaLight(v)
            m.emitter = a.radiance; m.emitter_weight = a.sampling_weight; m.emitter_light = a
        elif isinstance(v, dict) and 'type' in v and k not in ('to_world',):
            if _plugin_kind(v['type']) == 'emitter':     # Shape::initialize -> Emitter::set_shape: only surface emitters attach to a shape (point.cpp / spot.cpp / ... have no set_shape use)
                raise RuntimeError("Plugin \"%s\" is not a surface emitter: only `area` can be the child of a shape" % v['type'])
    return m


def _mk_scene(props, named, key):
    _check_props('scene', props, (), children=_ALL_KINDS)                 # Scene(props) (scene.cpp:26-70) walks props.objects(): children under any name, no other property
    children = {}
    for k, v in props.items():
        if k == 'type':
            continue
        if isinstance(v, dict) and v.get('type') in _BSDF_PLUGINS:
            named[k] = _create(v, named, k); children[k] = named[k]
    for k, v in props.items():
        if k == 'type' or k in children:
            continue
        obj = _resolve(v, named, k) if isinstance(v, dict) else v
        if isinstance(obj, (Mesh, Sensor, Integrator, BSDF, ShapeGroup, Instance, ConstantEmitter, EnvmapEmitter, PointLight, SpotLight, DirectionalEmitter)):
            if isinstance(obj, ShapeGroup):
                named[k] = obj
            children[k] = obj
    return Scene(children)


def _mk_shapegroup(props, named, key):
    _check_props('shapegroup', props, (), children=('shape',))            # ShapeGroup(props) (src/shapes/shapegroup.cpp): the child shapes, nothing else
    shapes = []; keys = []
    for k, v in props.items():
        if isinstance(v, dict) and 'type' in v:
            obj = _resolve(v, named, k)
            if isinstance(obj, Mesh):
                shapes.append(obj); keys.append(k)
        elif isinstance(v, Mesh):
            shapes.append(v); keys.append(k)
    g = ShapeGroup(shapes, keys)
    if key:
        named[key] = g
    return g


def _mk_instance(props, named, key):
    _check_props('instance', props, ('to_world',), children=('shape',))   # Instance(props) (src/shapes/instance.cpp:60-77)
    group = None
    for k, v in props.items():
        obj = _resolve(v, named, k) if isinstance(v, dict) else v
        if isinstance(obj, ShapeGroup):
            if group is not None:
                raise RuntimeError("Only a single shapegroup can be specified per instance.")
            group = obj
    if group is None:
        raise RuntimeError("A reference to a 'shapegroup' must be specified!")
    return Instance(group, props.get('to_world', ScalarTransform4f()))


def _mk_ply(props, named, key):
    _check_props('ply', props, _SHAPE_PROPS + ('filename', 'flip_tex_coords'), children=_SHAPE_CHILDREN)      # ply.cpp:113-118 over Mesh(props)
    if 'filename' not in props:
        raise RuntimeError("ply: the `filename` parameter is required")
    m = Mesh(key or "ply").from_ply(props['filename'], props.get('face_normals', False), props.get('flip_tex_coords', False),
                                    props.get('to_world'), props.get('flip_normals', False))
    return _shape_common(m, props, named)


def _mk_obj(props, named, key):
    _check_props('obj', props, _SHAPE_PROPS + ('filename', 'flip_tex_coords'), children=_SHAPE_CHILDREN)      # obj.cpp:98-113
    if 'filename' not in props:
        raise RuntimeError("obj: the `filename` parameter is required")
    m = Mesh(key or "obj").from_obj(props['filename'], props.get('face_normals', False), props.get('flip_tex_coords', True),
                                    props.get('to_world'), props.get('flip_normals', False))
    return _shape_common(m, props, named)


def _mk_serialized(props, named, key):
    _check_props('serialized', props, _SHAPE_PROPS + ('filename', 'shape_index'), children=_SHAPE_CHILDREN)   # serialized.cpp:225-244
    if 'filename' not in props:
        raise RuntimeError("serialized: the `filename` parameter is required")
    m = Mesh(key or "serialized").from_serialized(props['filename'], props.get('shape_index', 0), props.get('face_normals', None),
                                                  props.get('to_world'), props.get('flip_normals', False))
    return _shape_common(m, props, named)


def _mk_mesh(props, named, key):
    # not a reference plugin: the dict form of mi.Mesh(...) + params (faces / positions / normals / texcoords arrays)
    _check_props('mesh', props, ('faces', 'positions', 'normals', 'texcoords', 'to_world', 'silhouette_sampling_weight'), children=_SHAPE_CHILDREN)
    m = Mesh(key or "mesh").from_fields(props['faces'], props['positions'], props.get('normals'), props.get('texcoords'))
    if 'to_world' in props:
        m.transform(props['to_world'])
    return _shape_common(m, props, named)


for _name, _fn in {
    'scene': _mk_scene,
    'path': lambda p, n, k: Integrator(p),
    'prb': lambda p, n, k: Integrator(p),
    'perspective': lambda p, n, k: Sensor({kk: (_resolve(v, n, kk) if isinstance(v, dict) and 'type' in v else v) for kk, v in p.items()}),      # EVERY child object through the registry
    'orthographic': lambda p, n, k: Sensor({kk: (_resolve(v, n, kk) if isinstance(v, dict) and 'type' in v else v) for kk, v in p.items()}),
    'hdrfilm': lambda p, n, k: Film(p),
    'independent': lambda p, n, k: Sampler(p),
    'diffuse': lambda p, n, k: BSDF(p, id=k), 'dielectric': lambda p, n, k: BSDF(p, id=k), 'roughconductor': lambda p, n, k: BSDF(p, id=k),
    'roughplastic': lambda p, n, k: BSDF(p, id=k), 'conductor': lambda p, n, k: BSDF(p, id=k), 'plastic': lambda p, n, k: BSDF(p, id=k), 'twosided': _mk_twosided, 'constant': lambda p, n, k: ConstantEmitter(p), 'envmap': lambda p, n, k: EnvmapEmitter(p), 'point': lambda p, n, k: PointLight(p), 'spot': lambda p, n, k: SpotLight(p), 'directional': lambda p, n, k: DirectionalEmitter(p),
    'rectangle': lambda p, n, k: _shape_common(_rectangle(p), p, n),
    'cube': lambda p, n, k: _shape_common(_cube(p), p, n),
    'mesh': _mk_mesh, 'ply': _mk_ply, 'obj': _mk_obj, 'serialized': _mk_serialized,
    'shapegroup': _mk_shapegroup,
    'instance': _mk_instance,
    'gaussian': lambda p, n, k: p, 'box': lambda p, n, k: p, 'rgb': lambda p, n, k: p, 'bitmap': lambda p, n, k: p, 'area': lambda p, n, k: p,
}.items():
    register_plugin(_name, VARIANT, _fn)


def register_integrator(name, fn):
    """mi.register_integrator (src/render/python/scene_v.cpp:166-171)."""
    register_plugin(name, VARIANT, lambda p, n, k: fn(p))


def load_dict(d):
    """mi.load_dict (src/core/python/parser.cpp:561)."""
    if _variant is None:
        set_variant(VARIANT)
    return _create(d, {}, None)


# ---------------------------------------------------------------------------
#  mi.render + _RenderOp (src/python/python/util.py:344-528) as a torch autograd Function
# ---------------------------------------------------------------------------

def sample_tea_32(v0, v1, rounds=4):
    v0 &= 0xffffffff; v1 &= 0xffffffff; s = 0
    for _ in range(rounds):
        s = (s + 0x9e3779b9) & 0xffffffff
        v0 = (v0 + ((((v1 << 4) & 0xffffffff) + 0xa341316c) ^ ((v1 + s) & 0xffffffff) ^ ((v1 >> 5) + 0xc8013ea4))) & 0xffffffff
        v1 = (v1 + ((((v0 << 4) & 0xffffffff) + 0xad90777d) ^ ((v0 + s) & 0xffffffff) ^ ((v0 >> 5) + 0x7e95761e))) & 0xffffffff
    return v0, v1


def render(scene, params=None, sensor=0, integrator=None, seed=0, seed_grad=0, spp=0, spp_grad=0):
    torch = _torch()
    if integrator is None:
        integrator = scene.integrator()
    if integrator is None:
        raise Exception('No integrator specified! Add an integrator in the scene description or provide an integrator directly as argument.')
    if isinstance(sensor, int):
        if len(scene.sensors()) == 0:
            raise Exception('No sensor specified! Add a sensor in the scene description or provide a sensor directly as argument.')
        sensor = scene.sensors()[sensor]
    if spp_grad == 0:
        spp_grad = spp
    if seed_grad == 0:
        seed_grad = sample_tea_32(seed, 1)[0]
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different to ensure unbiased gradient computation!')
    if _variant == SCALAR_VARIANT:
        if any(getattr(v, 'requires_grad', False) for v in (params or {}).values()):
            raise RuntimeError("scalar_rgb renders are not differentiable (the reference's scalar variants have no AD either); use hip_ad_rgb")
        return _render_scalar(scene, integrator, sensor, seed, spp)
    keys = [k for k, v in (params or {}).items() if getattr(v, 'requires_grad', False)]
    fixed = [k for k in keys if k in scene._pose_keys() and scene._pose_keys()[k][0] != "emitter_tex"]      # (a light's radiance bitmap IS differentiable: area.cpp:64-70)
    if fixed:       # ParamFlags::NonDifferentiable in the reference's traverse(): dr.enable_grad on them has no effect there; here it is said
        raise RuntimeError("%s are not differentiable parameters in hip_ad_rgb (placement of sensors and delta emitters: ParamFlags::NonDifferentiable in the reference; "
                           "a spot light's cutoff_angle / beam_width: Differentiable there, updatable but without a gradient here)" % fixed)
    if not keys:
        return integrator.render(scene, sensor, seed, spp)
    params.update()
    # dr.enable_grad(params[key]) is what makes a parameter differentiable in the reference: the adjoint terms the requested keys need are switched
    # on for THIS call's backward pass only -- the caller's integrator keeps its own properties (optimising vertex positions first and alpha
    # second with one integrator must not leave `shape_gradients` on for the second render)
    overrides = {}
    if integrator.type == 'prb':
        shape_keys = [k for k in keys if k in scene._position_keys() or k in scene._instance_keys() or k in scene._rect_keys()]
        if shape_keys and integrator.shape_gradients is not True:
            overrides['shape_gradients'] = sorted(set(list(integrator.shape_gradients or [])) | set(shape_keys))
        if any(k in scene._bsdf_param_keys() for k in keys) and not integrator.bsdf_parameter_gradients:
            overrides['bsdf_parameter_gradients'] = True
        if any(scene._pose_keys().get(k, (None,))[0] == "emitter_tex" for k in keys) and not integrator.light_texel_gradients:
            overrides['light_texel_gradients'] = True
        if any(v[0] == "emit" for k, v in scene._param_keys().items() if k in keys) and not integrator.emitter_gradients:
            overrides['emitter_gradients'] = True

    class _RenderOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *tensors):
            # no synchronisation: the frame is enqueued on the current stream like any torch op (a scene whose BVH is too deep for the traversal stacks is refused
            # at creation, so there is no device-side error left to wait for); Integrator.render() called directly still evaluates
            return integrator.render(scene, sensor, seed, spp, evaluate=False)

        @staticmethod
        def backward(ctx, grad_out):
            saved = {k: getattr(integrator, k) for k in overrides}
            try:
                for k, v in overrides.items():
                    setattr(integrator, k, v)
                grads = integrator.render_backward(scene, params, grad_out, sensor, seed_grad, spp_grad)
            finally:
                for k, v in saved.items():
                    setattr(integrator, k, v)
            missing = [k for k in keys if k not in grads]
            if missing:       # e.g. emitter radiance with emitter_gradients=False, vertex positions without shape_gradients
                raise RuntimeError("mi.render(): the `prb` integrator cannot differentiate %s (integrator properties `emitter_gradients`, "
                                   "`shape_gradients`, `bsdf_parameter_gradients`, `light_texel_gradients`); differentiable keys of this render: %s" % (missing, sorted(grads)))
            return tuple(grads[k].reshape(params[k].shape) for k in keys)

    return _RenderOp.apply(*[params[k] for k in keys])
