"""Incremental updates of the acceleration data (row a5 `rebuild`; src/render/scene.cpp:517-540, src/render/scene_optix.inl:351-372) on the host arrays:
the refit runs the SAME HAR_HD code as the kernels of har_refit.hip (har_refit.h), the instance level is rebuilt by the same host code the library calls.
 * a refit of geometry that did not move reproduces the built nodes / triangle records bit for bit;
 * after vertices moved, ray queries through the refitted tree equal the brute-force intersector and a freshly built scene, bit for bit;
 * an instance update equals a fresh build (the TLAS build is deterministic: same hash);
 * the scene bounds of environment emitters follow; a mesh that carries an emitter asks for a new scene."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(O):
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    L.hh_scene_create.restype = C.c_void_p; L.hh_scene_destroy.argtypes = [C.c_void_p]
    L.hh_accel_hash.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.hh_scene_update_vertices.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    L.hh_scene_update_instances.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, O.c_f32p, O.c_f32p, C.c_char_p, C.c_int]
    L.hh_trace.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, O.c_f32p, O.c_f32p, C.c_int, C.c_int, O.c_f32p, O.c_f32p, O.c_f32p, O.c_u32p, O.c_u32p, O.c_u32p, C.c_void_p]
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    return L


def _create(L, scene):
    desc = scene.desc(); err = C.create_string_buffer(256)
    h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    return h


def _hash(L, h):
    out = (C.c_uint64 * 3)(); L.hh_accel_hash(h, out); return tuple(out)


def _rays(n, seed=1):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.9, 0.9, (3, n)).astype(np.float32); d = rng.normal(size=(3, n)).astype(np.float32); d /= np.linalg.norm(d, axis=0)
    return np.ascontiguousarray(o), np.ascontiguousarray(d.astype(np.float32)), np.full(n, 3.4e38, np.float32)


def _trace(L, O, h, o, d, maxt, naive=0):
    n = o.shape[1]
    t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    prim = np.zeros(n, np.uint32); shape = np.zeros(n, np.uint32); inst = np.zeros(n, np.uint32)
    p = lambda a: a.ctypes.data_as(O.c_u32p)
    assert L.hh_trace(h, n, O.fp(o), O.fp(d), O.fp(maxt), naive, 0, O.fp(t), O.fp(u), O.fp(v), p(prim), p(shape), p(inst), None) == 0
    return t, u, v, prim, shape, inst


def spheres(mi, flatten, grid=3, n_u=24, n_v=12, sky=False):
    d = mi.instanced_spheres_scene(width=16, height=16, spp=4, grid=grid, n_u=n_u, n_v=n_v, flatten=flatten)
    if sky:
        d.pop("ceiling"); d["sky"] = {"type": "constant", "radiance": {"type": "rgb", "value": [0.4, 0.5, 0.6]}}
    return d


def _move(V, rng, amount):
    W = V.copy(); W[:, :3] += rng.normal(scale=amount, size=(V.shape[0], 3)).astype(np.float32)
    return np.ascontiguousarray(W, np.float32)


@pytest.mark.parametrize("flatten", [True, False])
def test_refit_of_unmoved_geometry_is_the_built_tree(mi, O, flatten):
    L = _lib(O); scene = mi.load_dict(spheres(mi, flatten))
    h = _create(L, scene); before = _hash(L, h)
    err = C.create_string_buffer(256); area = C.c_double()
    keys = scene._position_keys()
    name = "ball004.positions" if flatten else "spheres.ball.positions"
    m = keys[name]
    assert L.hh_scene_update_vertices(h, m, O.fp(np.ascontiguousarray(scene.meshes[m]["V"], np.float32)), C.byref(area), err, 256) == 0, err.value
    assert _hash(L, h) == before and area.value > 0              # nodes, triangle records, instance records: bit for bit
    L.hh_scene_destroy(h)


@pytest.mark.parametrize("flatten", [True, False])
def test_refit_after_a_move_is_exact(mi, O, flatten):
    """moved vertices: the refitted tree answers ray queries like the brute-force loop over its own triangle records and like a freshly built scene"""
    L = _lib(O); rng = np.random.default_rng(3)
    d = spheres(mi, flatten); scene = mi.load_dict(d)
    h = _create(L, scene)
    name = "ball004.positions" if flatten else "spheres.ball.positions"
    m = scene._position_keys()[name]
    o, dd, maxt = _rays(20000)
    err = C.create_string_buffer(256)
    for step, amount in enumerate((0.002, 0.01, 0.05)):          # small, medium, large against a sphere radius of 0.08
        V = _move(scene.meshes[m]["V"], rng, amount)
        scene.meshes[m]["V"] = V
        assert L.hh_scene_update_vertices(h, m, O.fp(V), None, err, 256) == 0, err.value
        got = _trace(L, O, h, o, dd, maxt)
        brute = _trace(L, O, h, o, dd, maxt, naive=1)
        fresh_h = _create(L, scene); fresh = _trace(L, O, fresh_h, o, dd, maxt); L.hh_scene_destroy(fresh_h)
        assert np.isfinite(got[0]).sum() > 2000
        for a, b, c in zip(got, brute, fresh):
            assert np.array_equal(a, b) and np.array_equal(a, c), (step, amount)
    L.hh_scene_destroy(h)


def test_instance_update_equals_a_fresh_build(mi, O):
    L = _lib(O)
    d = spheres(mi, False, sky=True); scene = mi.load_dict(d)
    h = _create(L, scene)
    T = mi.ScalarTransform4f
    err = C.create_string_buffer(256)
    # move three instances: a single one, then a run of two
    for first, count in ((5, 1), (1, 2)):
        tw = []; to = []
        for k in range(count):
            t = T().translate([0.1 * (first + k) - 0.4, 0.3, -0.2]).rotate([0, 0, 1], 25.0 * (k + 1)).scale(1.3)
            i = first + k
            scene.instances[i] = (scene.instances[i][0], list(t.col_major_3x4()), list(t.inverse().col_major_3x4()))
            tw.append(np.asarray(scene.instances[i][1], np.float32)); to.append(np.asarray(scene.instances[i][2], np.float32))
        tw = np.ascontiguousarray(np.concatenate(tw)); to = np.ascontiguousarray(np.concatenate(to))
        assert L.hh_scene_update_instances(h, first, count, O.fp(tw), O.fp(to), err, 256) == 0, err.value
        fresh = _create(L, scene)
        assert _hash(L, h) == _hash(L, fresh)
        # ... and the environment emitter's bounding sphere followed: same picture
        sensor = scene.sensors()[0].har
        a = np.zeros((16, 16, 4), np.float32); b = np.zeros_like(a)
        assert L.hh_render(h, C.byref(sensor), 0, 3, 4, 5, 5, 0, 0, O.fp(a)) == 0 and L.hh_render(fresh, C.byref(sensor), 0, 3, 4, 5, 5, 0, 0, O.fp(b)) == 0
        assert np.array_equal(a, b) and a[..., :3].max() > 0
        L.hh_scene_destroy(fresh)
    assert L.hh_scene_update_instances(h, 8, 2, O.fp(tw), O.fp(to), err, 256) == 1 and b"out of bounds" in err.value
    L.hh_scene_destroy(h)


def test_vertex_update_of_an_instanced_mesh_moves_every_instance(mi, O):
    """a mesh inside a shape group: the group's BLAS is refitted, the boxes of all its instances and the TLAS follow -- render == fresh scene"""
    L = _lib(O); rng = np.random.default_rng(9)
    scene = mi.load_dict(spheres(mi, False, sky=True))
    h = _create(L, scene)
    m = scene._position_keys()["spheres.ball.positions"]
    V = scene.meshes[m]["V"].copy(); V[:, :3] *= 1.6; V = _move(V, rng, 0.004)       # grow the sphere: every instance box must grow
    scene.meshes[m]["V"] = V
    err = C.create_string_buffer(256)
    assert L.hh_scene_update_vertices(h, m, O.fp(V), None, err, 256) == 0, err.value
    fresh = _create(L, scene)
    hh, hf = _hash(L, h), _hash(L, fresh)
    assert hh[2] == hf[2]                                        # instance records (TLAS leaves): identical
    sensor = scene.sensors()[0].har
    a = np.zeros((16, 16, 4), np.float32); b = np.zeros_like(a)
    assert L.hh_render(h, C.byref(sensor), 0, 3, 8, 5, 5, 0, 0, O.fp(a)) == 0 and L.hh_render(fresh, C.byref(sensor), 0, 3, 8, 5, 5, 0, 0, O.fp(b)) == 0
    assert np.array_equal(a, b)
    L.hh_scene_destroy(fresh); L.hh_scene_destroy(h)


def test_emitter_meshes_ask_for_a_new_scene(mi, O):
    L = _lib(O); scene = mi.load_dict(mi.cornell_box())
    h = _create(L, scene)
    m = scene._position_keys()["light.positions"]
    err = C.create_string_buffer(256)
    assert L.hh_scene_update_vertices(h, m, O.fp(np.ascontiguousarray(scene.meshes[m]["V"], np.float32)), None, err, 256) == 2 and b"emitter" in err.value
    L.hh_scene_destroy(h)


def test_refit_area_grows_when_the_tree_degrades(mi, O):
    """the figure har_scene_update_vertices watches: shuffling vertices far from where the tree was built for inflates the sum of the node areas"""
    L = _lib(O); rng = np.random.default_rng(2)
    scene = mi.load_dict(spheres(mi, True, grid=2, n_u=32, n_v=16))
    h = _create(L, scene); m = scene._position_keys()["ball001.positions"]
    err = C.create_string_buffer(256); a0 = C.c_double(); a1 = C.c_double()
    V = np.ascontiguousarray(scene.meshes[m]["V"], np.float32)
    assert L.hh_scene_update_vertices(h, m, O.fp(V), C.byref(a0), err, 256) == 0
    W = V.copy(); W[:, :3] = W[rng.permutation(V.shape[0]), :3]
    assert L.hh_scene_update_vertices(h, m, O.fp(np.ascontiguousarray(W)), C.byref(a1), err, 256) == 0
    assert a1.value > 1.15 * a0.value             # one of four spheres (and the walls) of this BLAS
    L.hh_scene_destroy(h)


def _lib_positions(O):
    L = _lib(O)
    L.hh_scene_update_positions.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    L.hh_scene_get_vertices.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p]
    return L


@pytest.mark.parametrize("flatten", [True, False])
def test_device_resident_update_code_equals_the_host_update_bit_for_bit(mi, O, flatten):
    """har_scene_update_vertices_device's per-element code (har_vertex_update.h: positions -> records, Mesh::compute_normals as a per-vertex gather in (face, corner)
    order, shading triangles) run on the host arrays == the host update (har_mesh_compute_normals' serial loop + har_scene_update_vertices): same vertex records,
    same accel hash, same ray answers.  The gather adds a vertex's corner terms in the order the serial loop reaches them, so the float32 sums are the same numbers."""
    import mitsuba3_amd as pkg
    L = _lib_positions(O); rng = np.random.default_rng(11)
    scene = mi.load_dict(spheres(mi, flatten))
    name = "ball004.positions" if flatten else "spheres.ball.positions"
    m = scene._position_keys()[name]
    assert scene.meshes[m]["flags"] & 1                                   # the bumpy spheres carry vertex normals
    V0 = np.ascontiguousarray(scene.meshes[m]["V"], np.float32); F = np.ascontiguousarray(scene.meshes[m]["F"])
    P = np.ascontiguousarray(_move(V0, rng, 0.004)[:, :3])
    err = C.create_string_buffer(256)
    # (a) the host update: positions + har_mesh_compute_normals on the host, then the refit
    Vh = V0.copy(); Vh[:, :3] = P
    assert pkg.lib().har_mesh_compute_normals(Vh.shape[0], O.fp(Vh), F.shape[0], F.ctypes.data_as(O.c_u32p)) == 0
    ha = _create(L, scene)
    assert L.hh_scene_update_vertices(ha, m, O.fp(Vh), None, err, 256) == 0, err.value
    # (b) the device-resident update's code
    hb = _create(L, scene); area = C.c_double()
    assert L.hh_scene_update_positions(hb, m, O.fp(P), C.byref(area), err, 256) == 0, err.value
    Vb = np.zeros_like(V0); assert L.hh_scene_get_vertices(hb, m, O.fp(Vb)) == 0
    assert np.array_equal(Vb.view(np.uint32), Vh.view(np.uint32))          # positions, regenerated normals, texcoords: bit for bit
    assert np.abs(np.linalg.norm(Vb[:, 3:6], axis=1) - 1).max() < 1e-5 and np.abs(Vb[:, 3:6] - V0[:, 3:6]).max() > 1e-4
    assert _hash(L, ha) == _hash(L, hb) and area.value > 0
    o, d, maxt = _rays(4000, seed=5)
    for x, y in zip(_trace(L, O, ha, o, d, maxt), _trace(L, O, hb, o, d, maxt)):
        assert np.array_equal(x, y)
    img_a = np.zeros((16, 16, 4), np.float32); img_b = np.zeros_like(img_a)           # RGBW film
    sensor = scene.sensors()[0]
    assert L.hh_render(ha, C.byref(sensor.har), 0, 0, 4, 6, 5, 0, 16 * 16 * 4, O.fp(img_a)) == 0
    assert L.hh_render(hb, C.byref(sensor.har), 0, 0, 4, 6, 5, 0, 16 * 16 * 4, O.fp(img_b)) == 0
    assert np.array_equal(img_a, img_b)                                    # shading reads the rewritten shading triangles
    L.hh_scene_destroy(ha); L.hh_scene_destroy(hb)


def test_device_resident_update_refuses_emitter_meshes(mi, O):
    L = _lib_positions(O); scene = mi.load_dict(mi.cornell_box())
    h = _create(L, scene); err = C.create_string_buffer(256)
    m = scene._position_keys()["light.positions"]
    P = np.ascontiguousarray(scene.meshes[m]["V"][:, :3], np.float32)
    assert L.hh_scene_update_positions(h, m, O.fp(P), None, err, 256) == 2 and b"emitter" in err.value
    L.hh_scene_destroy(h)


def test_instance_level_refit_equals_rebuild_when_nothing_moved_and_is_exact_after_a_move(mi, O):
    """an instanced mesh updated the device's way: BLAS refit + instance boxes from the vertices + TLAS refit (the topology of the instance level stays).  Unmoved vertices:
    the refitted TLAS is the built one bit for bit (the boxes are build_tlas's exact vertex bounds, the node arithmetic is the builder's).  After a move: ray queries equal
    the brute-force loop and the host path's REBUILT instance level bit for bit (boxes only prune)."""
    L = _lib_positions(O)
    L.hh_scene_update_positions_instanced.argtypes = [C.c_void_p, C.c_uint32, O.c_f32p, C.c_char_p, C.c_int]
    rng = np.random.default_rng(21)
    scene = mi.load_dict(spheres(mi, False))
    m = scene._position_keys()["spheres.ball.positions"]
    V0 = np.ascontiguousarray(scene.meshes[m]["V"], np.float32)
    err = C.create_string_buffer(256)
    h = _create(L, scene); before = _hash(L, h)
    # flat-shaded comparison needs the regenerated normals on both sides: first an update with the SAME positions through both paths
    assert L.hh_scene_update_positions_instanced(h, m, O.fp(np.ascontiguousarray(V0[:, :3])), err, 256) == 0, err.value
    assert _hash(L, h) == before                                              # nodes (BLAS + TLAS), triangle records, instance records
    P = np.ascontiguousarray((V0[:, :3] * np.float32(1.25) + rng.normal(scale=3e-4, size=(V0.shape[0], 3))).astype(np.float32))
    assert L.hh_scene_update_positions_instanced(h, m, O.fp(P), err, 256) == 0, err.value
    h2 = _create(L, scene)
    assert L.hh_scene_update_positions(h2, m, O.fp(P), None, err, 256) == 0, err.value      # host path: the instance level is REBUILT
    o, d, maxt = _rays(6000, seed=8)
    a = _trace(L, O, h, o, d, maxt); b = _trace(L, O, h2, o, d, maxt); c = _trace(L, O, h, o, d, maxt, naive=1)
    assert int(np.isfinite(a[0]).sum()) > 500
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    L.hh_scene_destroy(h); L.hh_scene_destroy(h2)
