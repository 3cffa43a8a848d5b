"""The placement parameters mi.traverse exposes as NON-differentiable (the reference's ParamFlags::NonDifferentiable entries): '<sensor>.to_world' (perspective.cpp:177,
orthographic.cpp:95), '<emitter>.position' (point.cpp:86), '<emitter>.to_world' (spot.cpp:117, directional.cpp:96).  params.update() must leave the scene exactly as if it had
been loaded with the new placement: the records handed to the C ABI (and to the oracle) are compared with those of a freshly loaded scene, and the oracle renders both."""
import ctypes as C

import numpy as np
import pytest


def scene_dict(mi, bulb=(0.3, 0.2, 0.1), spot_origin=(0.3, 0.9, 0.2), sun=(0.3, -0.2, -1.0), cam=(0.1, 0.05, 3.9), ortho=True):
    d = mi.cornell_box()
    film = dict(d["sensor"]["film"]); film["width"] = 24; film["height"] = 24
    T = mi.ScalarTransform4f().look_at(origin=list(cam), target=[0, 0, 0], up=[0, 1, 0])
    d["sensor"] = {"type": "orthographic" if ortho else "perspective", "near_clip": 0.01, "far_clip": 100.0, "film": film, "to_world": T}
    d["bulb"] = {"type": "point", "position": list(bulb), "intensity": {"type": "rgb", "value": [0.5, 0.4, 0.3]}}
    d["spot"] = {"type": "spot", "cutoff_angle": 40.0, "intensity": {"type": "rgb", "value": [3.0, 2.0, 1.0]},
                 "to_world": mi.ScalarTransform4f().look_at(origin=list(spot_origin), target=[-0.2, -1.0, 0.1], up=[0, 0, 1])}
    sd = np.asarray(sun, np.float64); sd /= np.linalg.norm(sd)
    up = np.cross(sd, [1.0, 0.0, 0.0]); up /= np.linalg.norm(up)
    d["sun"] = {"type": "directional", "irradiance": {"type": "rgb", "value": [1.0, 0.8, 0.6]},
                "to_world": mi.ScalarTransform4f().look_at(origin=[0, 0, 0], target=[float(x) for x in sd], up=[float(x) for x in up])}
    return d


def emitter_records(scene):
    return [(e.get("type", 0), np.asarray(e["to_world"], np.float32), np.asarray(e.get("to_local", [0] * 12), np.float32), np.asarray(e["radiance"], np.float32)) for e in scene.emitters]


@pytest.mark.parametrize("ortho", [True, False])
def test_update_equals_a_fresh_load(mi, O, ortho):
    import torch
    a = mi.load_dict(scene_dict(mi, ortho=ortho))
    params = mi.traverse(a)
    for k in ("sensor.to_world", "bulb.position", "spot.to_world", "sun.to_world"):
        assert k in params and not params[k].requires_grad, k
    new = dict(bulb=(-0.2, 0.4, 0.3), spot_origin=(-0.4, 0.8, 0.0), sun=(-0.2, -0.5, -1.0), cam=(-0.3, 0.2, 3.5))
    b = mi.load_dict(scene_dict(mi, ortho=ortho, **new))
    pb = mi.traverse(b)
    for k in ("sensor.to_world", "bulb.position", "spot.to_world", "sun.to_world"):
        params[k] = pb[k].clone()
    params.update()
    for (ta, wa, la, ra), (tb, wb, lb, rb) in zip(emitter_records(a), emitter_records(b)):
        assert ta == tb and np.array_equal(wa, wb) and np.array_equal(ra, rb)
        if ta in (5, 6):
            assert np.allclose(la, lb, atol=1e-6)                       # the inverse is recomputed from the matrix (numpy) instead of tracked through look_at
    sa, sb = a.sensors()[0].har, b.sensors()[0].har
    assert bytes(sa)[:128] != bytes(mi.load_dict(scene_dict(mi, ortho=ortho)).sensors()[0].har)[:128]
    assert np.allclose(np.asarray(sa.to_world), np.asarray(sb.to_world), atol=1e-6) and np.array_equal(np.asarray(sa.sample_to_camera), np.asarray(sb.sample_to_camera))
    oa, sena = O.scene_from_product(a); ob, senb = O.scene_from_product(b)
    ia, _ = oa.render_path(sena, seed=1, spp=4, max_depth=5); ib, _ = ob.render_path(senb, seed=1, spp=4, max_depth=5)
    assert np.abs(ib).max() > 0 and np.allclose(ia, ib, rtol=1e-4, atol=1e-6)
    # and it is a different picture from the one before the update
    o0, s0 = O.scene_from_product(mi.load_dict(scene_dict(mi, ortho=ortho)))
    i0, _ = o0.render_path(s0, seed=1, spp=4, max_depth=5)
    assert np.linalg.norm(i0 - ib) > 0.05 * np.linalg.norm(ib)


def test_placement_is_not_differentiable(mi):
    scene = mi.load_dict(scene_dict(mi))
    params = mi.traverse(scene)
    params["bulb.position"] = params["bulb.position"].detach().clone().requires_grad_()
    with pytest.raises(RuntimeError, match="not differentiable"):
        mi.render(scene, params, spp=1)
    p2 = mi.traverse(mi.load_dict(scene_dict(mi, ortho=False)))
    import torch
    p2["sensor.to_world"] = torch.diag(torch.tensor([2.0, 2.0, 2.0, 1.0]))
    with pytest.raises(RuntimeError, match="Scale factors"):            # perspective.cpp:143-146 holds for updates too
        p2.update()


def test_spot_cone_parameters_are_updatable(mi, O):
    """SpotLight::traverse exposes cutoff_angle / beam_width (spot.cpp:115-116): params.update() re-lowers the cone; an updated scene is a freshly loaded one"""
    import torch
    scene = mi.load_dict(scene_dict(mi))
    params = mi.traverse(scene)
    assert float(params["spot.cutoff_angle"]) == 40.0 and float(params["spot.beam_width"]) == 30.0
    params["spot.cutoff_angle"] = torch.tensor([25.0]); params["spot.beam_width"] = torch.tensor([10.0]); params.update()
    d = scene_dict(mi); d["spot"]["cutoff_angle"] = 25.0; d["spot"]["beam_width"] = 10.0
    fresh = mi.load_dict(d)
    a = [e for e in scene.emitters if e.get("type") == 5][0]; b = [e for e in fresh.emitters if e.get("type") == 5][0]
    assert [float(x) for x in a["normal"]] == [float(x) for x in b["normal"]]
    sa, sensor = O.scene_from_product(scene); sb, _ = O.scene_from_product(fresh)
    ia, _ = sa.render_path(sensor, seed=1, spp=4, max_depth=4, raw=True); ib, _ = sb.render_path(sensor, seed=1, spp=4, max_depth=4, raw=True)
    assert np.array_equal(ia, ib)
    params["spot.beam_width"] = torch.tensor([30.0])
    with pytest.raises(RuntimeError, match="cutoff_angle"):
        params.update()
