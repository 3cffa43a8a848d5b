"""ctypes binding of libhip_ad_rgb.so (the C ABI in include/hip_ad_rgb.h).

This is the only bridge between the Python host layer and the HIP product.
It fails loudly when the shared library is missing: there is no eager /
PyTorch / CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAR_LIB_PATH", os.path.join(_HERE, "libhip_ad_rgb.so"))   # override: A/B builds in tools/

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class HarMesh(C.Structure):
    _fields_ = [("vertex_ptr", f32p), ("index_ptr", u32p), ("vertex_count", C.c_uint32),
                ("face_count", C.c_uint32), ("bsdf", C.c_uint32), ("emitter", C.c_int32),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class HarShapeGroup(C.Structure):
    _fields_ = [("first_mesh", C.c_uint32), ("mesh_count", C.c_uint32)]


class HarInstance(C.Structure):
    _fields_ = [("group", C.c_uint32), ("to_world", C.c_float * 12), ("to_object", C.c_float * 12)]


class HarBSDF(C.Structure):
    _fields_ = [("type", C.c_uint32), ("texture", C.c_int32), ("reflectance", C.c_float * 3), ("flags", C.c_uint32),
                ("reflectance2", C.c_float * 3), ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("eta", C.c_float),
                ("eta_c", C.c_float * 3), ("k_c", C.c_float * 3), ("back", C.c_int32)]


class HarTexture(C.Structure):
    _fields_ = [("data", f32p), ("width", C.c_uint32), ("height", C.c_uint32), ("mode", C.c_uint32), ("reserved", C.c_uint32), ("to_uv", C.c_float * 6)]


class HarEmitter(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mesh", C.c_uint32), ("radiance", C.c_float * 3),
                ("to_world", C.c_float * 12), ("normal", C.c_float * 3), ("inv_area", C.c_float), ("to_local", C.c_float * 12), ("sampling_weight", C.c_float), ("radiance_texture", C.c_uint32)]


class HarMeshData(C.Structure):
    _fields_ = [("vertices", f32p), ("faces", u32p), ("vertex_count", C.c_uint32), ("face_count", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class HarImage(C.Structure):
    _fields_ = [("data", f32p), ("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("reserved", C.c_uint32)]


class HarSceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(HarMesh)), ("mesh_count", C.c_uint32), ("top_mesh_count", C.c_uint32),
                ("groups", C.POINTER(HarShapeGroup)), ("group_count", C.c_uint32), ("pad0", C.c_uint32),
                ("instances", C.POINTER(HarInstance)), ("instance_count", C.c_uint32), ("pad1", C.c_uint32),
                ("bsdfs", C.POINTER(HarBSDF)), ("bsdf_count", C.c_uint32), ("pad2", C.c_uint32),
                ("textures", C.POINTER(HarTexture)), ("texture_count", C.c_uint32), ("pad3", C.c_uint32),
                ("emitters", C.POINTER(HarEmitter)), ("emitter_count", C.c_uint32), ("pad4", C.c_uint32)]


class HarSensor(C.Structure):
    _fields_ = [("sample_to_camera", C.c_float * 16), ("to_world", C.c_float * 16),
                ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_width", C.c_uint32), ("film_height", C.c_uint32),
                ("crop_offset_x", C.c_uint32), ("crop_offset_y", C.c_uint32),
                ("crop_width", C.c_uint32), ("crop_height", C.c_uint32),
                ("rfilter", C.c_uint32), ("rfilter_stddev", C.c_float), ("rfilter_param1", C.c_float),
                ("sample_border", C.c_uint32), ("principal_point_offset_x", C.c_float), ("principal_point_offset_y", C.c_float),
                ("projection", C.c_uint32)]


class HarBSDFContext(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("type_mask", C.c_uint32), ("component", C.c_uint32)]


class HarStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("vertices", C.c_uint64), ("closest_rays", C.c_uint64),
                ("shadow_rays", C.c_uint64)]


# every symbol include/hip_ad_rgb.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "har_last_error": (C.c_char_p, []),
    "har_device_arch": (C.c_char_p, []),
    "har_scene_create": (C.c_int, [C.POINTER(HarSceneDesc), C.POINTER(vp)]),
    "har_scene_destroy": (C.c_int, [vp]),
    "har_scene_set_reflectance": (C.c_int, [vp, C.c_uint32, f32p]),
    "har_scene_set_emitter_radiance": (C.c_int, [vp, C.c_uint32, f32p]),
    "har_integrator_set_grad_emitters": (C.c_int, [vp, vp]),
    "har_integrator_set_grad_bsdf_params": (C.c_int, [vp, vp]),
    "har_integrator_set_grad_light_texels": (C.c_int, [vp, C.c_int]),
    "har_integrator_set_grad_positions": (C.c_int, [vp, vp, vp]),
    "har_integrator_set_grad_instances": (C.c_int, [vp, vp, vp]),
    "har_integrator_set_hide_emitters": (C.c_int, [vp, C.c_int]),
    "har_integrator_set_alpha_film": (C.c_int, [vp, vp]),
    "har_integrator_set_film_window": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "har_scene_set_texture": (C.c_int, [vp, C.c_uint32, f32p]),
    "har_scene_set_bsdf_params": (C.c_int, [vp, C.c_uint32, C.c_void_p]),
    "har_scene_set_delta_emitter": (C.c_int, [vp, C.c_uint32, C.c_void_p]),
    "har_scene_accel_info": (C.c_int, [vp, u64p]),
    "har_scene_sample_emitter": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "har_scene_pdf_emitter": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp]),
    "har_scene_set_emitter_sampling_weights": (C.c_int, [vp, f32p, C.c_uint32]),
    "har_scene_set_texture_to_uv": (C.c_int, [vp, C.c_uint32, f32p]),
    "har_scene_update_instances": (C.c_int, [vp, C.c_uint32, C.c_uint32, f32p, f32p, vp]),
    "har_scene_update_vertices": (C.c_int, [vp, C.c_uint32, f32p, vp]),
    "har_scene_update_vertices_device": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "har_scene_get_vertices": (C.c_int, [vp, C.c_uint32, f32p, vp]),
    "har_scene_update_instances_device": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp]),
    "har_scene_get_instances": (C.c_int, [vp, C.c_uint32, C.c_uint32, f32p, f32p, vp]),
    "har_scene_refit_info": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "har_multi_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]),
    "har_multi_destroy": (C.c_int, [vp]),
    "har_multi_replica": (C.c_int, [vp, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]),
    "har_multi_render": (C.c_int, [vp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]),
    "har_multi_render_backward": (C.c_int, [vp, C.c_void_p, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(vp), vp, vp]),
    "har_band_rebalance": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "har_multi_info": (C.c_int, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_char_p, C.c_uint32]),
    "har_scene_set_texture_device": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "har_scene_set_reflectance_device": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "har_scene_set_emitter_radiance_device": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "har_ray_intersect_preliminary": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "har_ray_test": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, C.c_int, vp, vp]),
    "har_compute_surface_interaction": (C.c_int, [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp]),
    "har_ray_intersect": (C.c_int, [vp, C.c_uint32, vp, vp, vp, C.c_uint32, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "har_sampler_seed": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]),
    "har_sampler_next_1d": (C.c_int, [C.c_uint32, vp, vp, vp, vp, vp]),
    "har_sampler_next_2d": (C.c_int, [C.c_uint32, vp, vp, vp, vp, vp]),
    "har_bsdf_eval_pdf": (C.c_int, [vp, C.c_uint32, C.POINTER(HarBSDFContext), C.c_uint32, vp, vp, vp, vp, vp, vp, vp]),
    "har_bsdf_eval": (C.c_int, [vp, C.c_uint32, C.POINTER(HarBSDFContext), C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "har_bsdf_pdf": (C.c_int, [vp, C.c_uint32, C.POINTER(HarBSDFContext), C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "har_bsdf_sample": (C.c_int, [vp, C.c_uint32, C.POINTER(HarBSDFContext), C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "har_image_write_exr": (C.c_int, [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "har_image_write_pfm": (C.c_int, [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "har_image_read": (C.c_int, [C.c_char_p, vp]),
    "har_image_free": (None, [vp]),
    "har_integrator_set_replay_cache": (C.c_int, [vp, C.c_int]),
    "har_integrator_set_material_queues": (C.c_int, [vp, C.c_int]),
    "har_integrator_set_packet_tracing": (C.c_int, [vp, C.c_int]),
    "har_mesh_load_ply": (C.c_int, [C.c_char_p, C.c_int, C.c_int, f32p, C.c_int, vp]),
    "har_mesh_load_obj": (C.c_int, [C.c_char_p, C.c_int, C.c_int, f32p, C.c_int, vp]),
    "har_mesh_load_serialized": (C.c_int, [C.c_char_p, C.c_int, C.c_int, f32p, C.c_int, vp]),
    "har_mesh_compute_normals": (C.c_int, [C.c_uint32, vp, C.c_uint32, vp]),
    "har_mesh_free": (None, [vp]),
    "har_sensor_sample_ray": (C.c_int, [C.POINTER(HarSensor), C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "har_film_put": (C.c_int, [C.POINTER(HarSensor), C.c_uint32, vp, vp, vp, vp, vp]),
    "har_film_develop": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp]),
    "har_film_develop_format": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp]),
    "har_set_allocator": (C.c_int, [vp, vp, vp]),
    "har_integrator_create": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(vp)]),
    "har_integrator_destroy": (C.c_int, [vp]),
    "har_render": (C.c_int, [vp, vp, C.POINTER(HarSensor), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]),
    "har_integrator_set_samples_per_pass": (C.c_int, [vp, C.c_uint32]),
    "har_render_pass_layout": (C.c_int, [vp, C.POINTER(HarSensor), C.c_uint32, u32p, u32p]),
    "har_render_weights": (C.c_int, [C.POINTER(HarSensor), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]),
    "har_render_backward": (C.c_int, [vp, vp, C.POINTER(HarSensor), vp, vp, C.c_uint32, C.c_uint32, C.c_uint64,
                                      C.c_uint64, vp, C.POINTER(vp), vp]),
    "har_render_scalar": (C.c_int, [C.POINTER(HarSceneDesc), C.POINTER(HarSensor), C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, vp, u32p]),
    "har_render_forward": (C.c_int, [vp, vp, C.POINTER(HarSensor), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp]),
    "har_integrator_sample": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "har_sampler_clone": (C.c_int, [C.c_uint32, vp, vp, vp, vp, vp]),
    "har_sampler_advance": (C.c_int, [C.c_uint32, vp, vp, vp]),
    "har_render_stats": (C.c_int, [vp, C.POINTER(HarStats)]),
    "har_integrator_set_profiling": (C.c_int, [vp, C.c_int]),
    "har_render_timing": (C.c_int, [vp, f32p, u32p]),
    "har_transform_translate": (C.c_int, [f32p, f32p]),
    "har_transform_scale": (C.c_int, [f32p, f32p]),
    "har_transform_rotate": (C.c_int, [f32p, C.c_float, f32p]),
    "har_transform_look_at": (C.c_int, [f32p, f32p, f32p, f32p]),
    "har_transform_mul": (C.c_int, [f32p, f32p, f32p]),
    "har_transform_inverse": (C.c_int, [f32p, f32p]),
    "har_orthographic_sensor": (C.c_int, [f32p, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                          C.POINTER(HarSensor)]),
    "har_perspective_sensor": (C.c_int, [f32p, C.c_double, C.c_char_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                         C.POINTER(HarSensor)]),
    "har_shape_rectangle": (C.c_int, [f32p, C.c_int, f32p, u32p, f32p, f32p]),
    "har_shape_cube": (C.c_int, [f32p, f32p, u32p]),
    "har_mesh_transform": (C.c_int, [f32p, C.c_uint32, f32p, C.c_uint32, u32p, C.c_int]),
}

_LIB = None


class HarError(RuntimeError):
    pass


def lib():
    """Load libhip_ad_rgb.so; raise (never fall back) if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HarError("libhip_ad_rgb.so is missing: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


# --- device memory through PyTorch's caching allocator (har_set_allocator): the library's workspaces (tens of GB for a 2^26-lane wavefront) and scene arrays then
# live in the process's one pool -- visible in torch.cuda.memory_allocated(), returned to it when an integrator goes away, never competing with it for the
# device.  HAR_TORCH_ALLOCATOR=0: plain hipMalloc / hipFree.
_ALLOC_CB = None; _FREE_CB = None; _ALLOC_STATE = {"installed": False, "closing": False}


def install_torch_allocator():
    global _ALLOC_CB, _FREE_CB
    if _ALLOC_STATE["installed"] or os.environ.get("HAR_TORCH_ALLOCATOR", "1") == "0" or os.environ.get("HAR_DEBUG_GUARD"):
        return
    import atexit
    import torch
    if not torch.cuda.is_available():
        return

    def _alloc(nbytes, user):
        try:
            return torch.cuda.caching_allocator_alloc(int(nbytes), torch.cuda.current_device(), torch.cuda.current_stream())
        except torch.cuda.OutOfMemoryError:          # the library reports "out of memory" through its own error path (and falls back to a smaller workspace)
            return None
        except Exception as e:                       # anything else is not an out-of-memory condition: say so, then fail the allocation
            import sys
            sys.stderr.write("[mitsuba3_amd] torch.cuda.caching_allocator_alloc(%d) raised %r\n" % (int(nbytes), e))
            return None

    def _free(ptr, user):
        if ptr and not _ALLOC_STATE["closing"]:
            try:
                torch.cuda.caching_allocator_delete(ptr)
            except Exception:
                pass

    _ALLOC_CB = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)(_alloc)
    _FREE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)(_free)
    check(lib().har_set_allocator(C.cast(_ALLOC_CB, C.c_void_p), C.cast(_FREE_CB, C.c_void_p), None))
    _ALLOC_STATE["installed"] = True
    # objects that outlive the interpreter's orderly phase free their blocks while torch is being torn down: leave those to the process exit
    atexit.register(lambda: _ALLOC_STATE.__setitem__("closing", True))


def check(rc):
    if rc != 0:
        msg = lib().har_last_error()
        raise HarError(msg.decode() if msg else "hip_ad_rgb call failed (rc=%d)" % rc)
