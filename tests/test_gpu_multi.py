"""Single-call multi-GPU entry of the C ABI (har_multi_*: one host thread, a scene replica + integrator + stream per device, row bands with global lane indices,
ONE reduce of the film on devices[0]; the reference's contract is one Integrator::render call from one thread, integrator.h:74-79).
On a one-GPU box: a group of one device equals har_render; a group that names the device twice / three times runs the whole band + reduce machinery on it (the
collective then is the peer-copy + add path: one physical device cannot form an RCCL communicator).  The RCCL branch needs two devices and is skipped otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))


def _scene(mi, res=64, spp=16):
    return mi.load_dict(mi.instanced_spheres_scene(width=res, height=res, spp=spp, grid=3, n_u=24, n_v=12))


def test_group_of_one_device_is_har_render(mi, O):
    scene = _scene(mi)
    single = mi.render(scene, spp=16, seed=3).cpu().numpy()
    st1 = scene.integrator().stats()
    g = mi.DeviceGroup(scene, devices=[0])
    img = g.render(spp=16, seed=3).cpu().numpy()
    assert rel_l2(img, single) < 1e-6 and g.stats() == st1                      # same samples: same counters; the film's float atomics have no fixed order
    assert g.info()["reduce"].startswith("single device")
    osc, sensor = O.scene_from_product(scene)
    ref, ost = osc.render_path(sensor, seed=3, spp=16, max_depth=scene.integrator().max_depth, rr_depth=scene.integrator().rr_depth)
    assert rel_l2(img, ref) < 1e-4 and g.stats()["vertices"] == ost.vertices
    film = g.render(spp=16, seed=3, develop=False)
    want = scene.integrator().render_film(scene, 0, 3, 16)                     # RGBW accumulation, not developed
    assert tuple(film.shape) == (64, 64, 4) and rel_l2(film.cpu().numpy(), want.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("n", [2, 3])
def test_bands_on_replicas_sum_to_the_single_render(mi, n):
    scene = _scene(mi)
    single = mi.render(scene, spp=16, seed=5).cpu().numpy()
    st1 = scene.integrator().stats()
    g = mi.DeviceGroup(scene, devices=[0] * n)
    for frame in range(5):                                                      # the bands are re-cut from the measured times of the first frames
        img = g.render(spp=16, seed=5).cpu().numpy()
        assert rel_l2(img, single) < 1e-6, frame
        assert g.stats() == st1, frame                                          # the union of the bands draws exactly the single render's samples
    info = g.info()
    assert "peer copies" in info["reduce"] and info["band_rows"][0] == 0 and info["band_rows"][-1] == 64
    assert all(b > a for a, b in zip(info["band_rows"], info["band_rows"][1:])) and all(t > 0 for t in info["band_ms"])


def test_replica_handles_take_parameter_updates(mi):
    import ctypes as C
    import torch
    scene = _scene(mi, res=32)
    g = mi.DeviceGroup(scene, devices=[0, 0])
    before = g.render(spp=8, seed=1).cpu().numpy()
    params = mi.traverse(scene)
    key = next(k for k in params.keys() if k.endswith("reflectance.value"))
    kind, b = scene._param_keys()[key]
    rgb = torch.tensor([0.9, 0.1, 0.1], device="cuda")
    for k in range(2):
        sc, _, dev = g.replica(k)
        assert dev == 0
        assert mi.lib().har_scene_set_reflectance_device(sc, b.index, rgb.data_ptr(), None) == 0
    torch.cuda.synchronize()
    after = g.render(spp=8, seed=1).cpu().numpy()
    params[key] = rgb; params.update()
    want = mi.render(scene, spp=8, seed=1).cpu().numpy()
    assert rel_l2(after, want) < 1e-6 and rel_l2(after, before) > 1e-3


def test_two_devices_reduce_over_rccl(mi):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the RCCL branch of har_multi_render: ncclCommInitAll + one ncclReduce)")
    scene = _scene(mi)
    single = mi.render(scene, spp=16, seed=7).cpu().numpy()
    g = mi.DeviceGroup(scene, devices=[0, 1])
    assert "ncclReduce" in g.info()["reduce"]
    for _ in range(4):
        assert rel_l2(g.render(spp=16, seed=7).cpu().numpy(), single) < 1e-6
    assert g.stats() == scene.integrator().stats()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_group_render_backward_equals_the_single_device_adjoint(mi, n):
    """har_multi_render_backward: weights of every band -> one all-reduce -> every replica replays its band -> one reduce of ONE flat gradient buffer per device;
    gradients w.r.t. bitmap texels, constant albedos and the emitter's radiance equal integrator.render_backward of the single scene (float atomics: 1e-4)"""
    import torch
    d = mi.textured_cornell_box(res=48, tex_res=32, spp=16)
    d["integrator"]["emitter_gradients"] = True
    scene = mi.load_dict(d)
    integ = scene.integrator()
    torch.manual_seed(3)
    grad_in = torch.rand((48, 48, 3), device="cuda") / (48 * 48 * 3)
    want = integ.render_backward(scene, None, grad_in, seed=11, spp=16)
    g = mi.DeviceGroup(scene, devices=[0] * n)
    for frame in range(4 if n > 1 else 1):                      # the adjoint's bands are re-cut from the first calls' device times
        got = g.render_backward(grad_in, seed=11, spp=16)
        assert set(got) == set(want)
        for k in want:
            a, b = got[k].cpu().numpy(), want[k].cpu().numpy()
            assert np.isfinite(a).all() and np.abs(b).max() > 0, k
            assert rel_l2(a, b) < 1e-4, (k, frame, rel_l2(a, b))
    # the forward render of the same group still works between adjoint calls (separate bands, shared replicas)
    img = g.render(spp=16, seed=2).cpu().numpy()
    assert rel_l2(img, mi.render(scene, spp=16, seed=2).detach().cpu().numpy()) < 1e-6


def test_bands_adapt_in_a_loop_that_never_synchronises(mi):
    """a host loop runs ahead of the device: the frame before the one being enqueued has usually not finished, so the bands are re-cut from the newest frame that HAS
    (a ring of four measured frames, csrc/har_multi.hip) -- two dozen frames enqueued back to back must have adapted by the end"""
    import torch
    scene = _scene(mi, res=128, spp=64)
    g = mi.DeviceGroup(scene, devices=[0, 0, 0])
    imgs = [g.render(spp=64, seed=9) for _ in range(24)]             # no .cpu(), no synchronize between the calls: the host may be two dozen frames ahead
    torch.cuda.synchronize()
    imgs += [g.render(spp=64, seed=9) for _ in range(3)]              # the measured frames of the burst have finished by now: the next call re-cuts from them
    torch.cuda.synchronize()
    info = g.info()
    assert all(t > 0 for t in info["band_ms"]), info                    # at least one re-cut happened from measured times
    single = mi.render(scene, spp=64, seed=9).cpu().numpy()
    for img in (imgs[0], imgs[-1]):
        assert rel_l2(img.cpu().numpy(), single) < 1e-6
