"""mitsuba3_amd -- MI355X-native `hip_ad_rgb` hot path for Mitsuba 3.

Forward `path` and Path-Replay-Backprop `prb` on triangle scenes as hand-written
HIP kernels for gfx950 behind a C ABI (include/hip_ad_rgb.h), with a Python host
layer that mirrors the reference's `mitsuba` module for this path:

    import mitsuba3_amd as mi
    mi.set_variant('hip_ad_rgb')
    scene = mi.load_dict(mi.cornell_box())
    image = mi.render(scene, spp=256)
"""
from ._capi import HarError, lib, LIB_PATH            # noqa: F401
from .core import (                                    # noqa: F401
    set_variant, variant, variants, ScalarTransform4f, Transform4f, ScalarTransform3f, Transform3f, AreaLight, cornell_box, load_dict, render, traverse,
    register_plugin, register_integrator, Scene, Sensor, Film, Sampler, BSDF, BSDFContext, BSDFFlags, TransportMode, RayFlags, Mesh, ShapeGroup, Instance,
    Integrator, Ray3f, PreliminaryIntersection3f, SurfaceInteraction3f, SceneParameters, develop_film, sample_tea_32,
    Bitmap, write_bitmap, ConstantEmitter, EnvmapEmitter, PointLight, SpotLight, DirectionalEmitter, DeviceGroup, ParamFlags,
)
from . import core                                     # noqa: F401
from .distributed import render_distributed, render_backward_distributed, lane_range   # noqa: F401
from .scenes import instanced_spheres_scene, textured_cornell_box                       # noqa: F401
from . import parser                                   # noqa: F401
from .parser import load_file, load_string             # noqa: F401
