"""The wave-shared ("packet") descent of the camera rays (k_trace_packet, har_integrator_set_packet_tracing) against the per-ray traversal kernel: the same
intersections bit for bit -- the packet's conservative box test only prunes less, the triangle test is the per-ray one (Mesh::moeller_trumbore, mesh.h:1130-1155) --
for coherent rays (a pixel's samples), for rays that share nothing (every packet falls back to the per-ray kernel through the list) and for renders."""
import numpy as np
import pytest

from tests.test_gpu_boundary import _scenes, rel_l2

pytestmark = pytest.mark.gpu


def _camera_like_rays(scene, mi, res, spp, seed):
    """spp jittered rays per pixel, pixel-major like the wavefront of render() (integrator.cpp:322-334)"""
    rng = np.random.default_rng(seed)
    ys, xs = np.divmod(np.repeat(np.arange(res * res), spp), res)
    pos = np.stack([(xs + rng.random(xs.size)) / res, (ys + rng.random(xs.size)) / res]).astype(np.float32)
    ray, _ = scene.sensors()[0].sample_ray(0.0, 0.0, pos)
    return ray


@pytest.mark.parametrize("kind", ["path", "prb"])
def test_packet_descent_equals_per_ray_traversal(mi, O, kind):
    for name, d in _scenes(mi, O):
        outs = {}
        for packet in (False, True):
            d["integrator"] = {"type": kind, "max_depth": 4, "rr_depth": 3, "packet_tracing": packet}
            scene = mi.load_dict(d)
            res, spp = 24, 64
            coherent = _camera_like_rays(scene, mi, res, spp, 3)
            n = len(coherent)
            rng = np.random.default_rng(9)
            o = rng.uniform(-0.8, 0.8, (3, n)).astype(np.float32); dd = rng.normal(size=(3, n)).astype(np.float32); dd /= np.linalg.norm(dd, axis=0)
            incoherent = mi.Ray3f(o, dd.astype(np.float32), np.full(n, 3.402823466e+38, np.float32))
            # a ragged wavefront (not a multiple of 64 rays) of half-coherent packets: 32 samples of one pixel next to 32 of another
            m = n - 37
            mixed = mi.Ray3f(coherent.o[:, :m], coherent.d[:, :m], coherent.maxt[:m])
            res_k = []
            for ray in (coherent, incoherent, mixed):
                sampler = mi.Sampler({"sample_count": 4, "seed": 1}); sampler.seed(7, len(ray))
                spec, valid = scene.integrator().sample(scene, sampler, ray)
                res_k.append((spec.cpu().numpy(), valid.cpu().numpy(), sampler.state.cpu().numpy().view(np.uint64).copy()))
            outs[packet] = res_k
        for a, b in zip(outs[False], outs[True]):
            assert np.array_equal(a[0], b[0]), (name, kind)          # per-lane radiance: no atomics anywhere on this path, bit-identical or wrong
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (name, kind)
        assert np.abs(outs[True][0][0]).max() > 0


def test_packet_descent_render_equals_per_ray_render(mi, O):
    """render() at 64 spp takes the packet path by itself; forcing it off gives the same paths (vertex / ray counters) and the same picture"""
    for name, d in _scenes(mi, O):
        imgs = {}; stats = {}
        for packet in (False, None):
            d["integrator"] = {"type": "path", "max_depth": 5, "packet_tracing": packet} if packet is not None else {"type": "path", "max_depth": 5}
            scene = mi.load_dict(d)
            imgs[packet] = mi.render(scene, spp=64, seed=4).cpu().numpy()
            stats[packet] = scene.integrator().stats()
        assert rel_l2(imgs[None], imgs[False]) < 1e-6, name
        assert stats[None] == stats[False], (name, stats)


def test_packet_descent_far_from_the_origin(mi):
    """the packet's box test rebuilds the child planes in world coordinates (fma(q, scale, origin)) before it subtracts the rays' origin bounds, the per-ray test keeps
    (origin - o) and q * scale apart: at coordinates of ~1e4 the two differ by ulp(1e4) ~ 1e-3 in the plane positions.  The boxes are padded by 2e-5 * |coordinate|
    (pad_box: 0.2 there, ~170 ulps), so the packet test still only prunes less -- checked here: same radiance bit for bit with the scene moved to (1e4, -2e4, 3e4)"""
    T = mi.ScalarTransform4f
    off = T().translate([1.0e4, -2.0e4, 3.0e4])
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 24; d["sensor"]["film"]["height"] = 24
    for k, v in d.items():
        if isinstance(v, dict) and v.get("type") in ("rectangle", "cube"):
            v["to_world"] = off @ v["to_world"]
    d["sensor"]["to_world"] = off @ d["sensor"]["to_world"]
    outs = {}
    for packet in (False, True):
        d["integrator"] = {"type": "path", "max_depth": 4, "rr_depth": 3, "packet_tracing": packet}
        scene = mi.load_dict(d)
        ray = _camera_like_rays(scene, mi, 24, 64, 3)
        sampler = mi.Sampler({"sample_count": 4, "seed": 1}); sampler.seed(7, len(ray))
        spec, valid = scene.integrator().sample(scene, sampler, ray)
        outs[packet] = (spec.cpu().numpy(), valid.cpu().numpy())
    assert np.array_equal(outs[False][0], outs[True][0]) and np.array_equal(outs[False][1], outs[True][1])
    assert outs[True][1].mean() > 0.9 and np.abs(outs[True][0]).max() > 0
