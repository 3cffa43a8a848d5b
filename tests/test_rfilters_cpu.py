"""Reconstruction filters (src/rfilters/*.cpp): the oracle against the reference's own spot checks (src/rfilters/tests/test_rfilter.py:8-53) and
closed forms, the product's HAR_HD code against the oracle, and a render through the host pipeline with each filter."""
import ctypes as C
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

FILTERS = {          # type id, parameter 0, parameter 1, radius, film dictionary
    "box": (0, 0.5, 0.0, 0.5, {"type": "box"}),
    "gaussian": (1, 0.5, 0.0, 2.0, {"type": "gaussian"}),
    "tent": (2, 1.0, 0.0, 1.0, {"type": "tent"}),
    "tent_wide": (2, 1.25, 0.0, 1.25, {"type": "tent", "radius": 1.25}),
    "mitchell": (3, 1 / 3, 1 / 3, 2.0, {"type": "mitchell"}),
    "mitchell_bc": (3, 0.1, 0.6, 2.0, {"type": "mitchell", "B": 0.1, "C": 0.6}),
    "catmullrom": (4, 0.0, 0.0, 2.0, {"type": "catmullrom"}),
    "lanczos": (5, 3.0, 0.0, 3.0, {"type": "lanczos"}),
    "lanczos2": (5, 2.0, 0.0, 2.0, {"type": "lanczos", "lobes": 2}),
}


def test_reference_spot_checks(O):
    """test_rfilter.py:8-53, same tolerances"""
    ev = lambda name, x: float(O.lib().orc_rfilter_eval2(FILTERS[name][0], FILTERS[name][1], FILTERS[name][2], float(x)))
    assert ev("box", 0.49) == 1 and ev("box", 0.51) == 0
    assert abs(ev("gaussian", 0.2) - 0.9227) < 8e-3 and ev("gaussian", 2.1) == 0
    assert abs(ev("lanczos", 1.4) - (-0.14668)) < 1e-2 and ev("lanczos", 3.1) == 0
    assert abs(ev("mitchell", 0) - 0.8888) < 1e-3 and ev("mitchell", 2.1) == 0
    assert abs(ev("catmullrom", 0) - 0.9765) < 5e-2 and ev("catmullrom", 2.1) == 0
    assert abs(ev("tent", 0.1) - 0.903) < 5e-2 and ev("tent", 1.1) == 0


def test_closed_forms_and_partition_of_unity(O):
    ev = lambda name, x: float(O.lib().orc_rfilter_eval2(FILTERS[name][0], FILTERS[name][1], FILTERS[name][2], float(x)))
    for x in np.linspace(-3.2, 3.2, 129):
        assert abs(ev("tent_wide", x) - max(0.0, 1 - abs(x) / 1.25)) < 1e-6
        sinc = lambda t: 1.0 if t == 0 else math.sin(math.pi * t) / (math.pi * t)
        want = sinc(x) * sinc(x / 3) if abs(x) <= 3 else 0.0
        assert abs(ev("lanczos", x) - want) < 2e-6, x
    # Mitchell-Netravali kernels (any B, C) and Catmull-Rom sum to one over the integer translates
    for name in ("mitchell", "mitchell_bc", "catmullrom", "tent"):
        for frac in (0.0, 0.13, 0.5, 0.77):
            assert abs(sum(ev(name, frac + k) for k in range(-3, 4)) - 1.0) < 1e-5, (name, frac)


@pytest.mark.parametrize("name", list(FILTERS))
def test_product_filter_code_and_film_weights_match_oracle(mi, O, name):
    """HAR_HD rfilter_eval (host build) == oracle on a dense grid; a Cornell render through the host pipeline splats with the same weights"""
    from tests.test_cpu_host import oracle_scene_from, rel_l2
    tid, p0, p1, radius, fd = FILTERS[name]
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 20; d["sensor"]["film"]["height"] = 20; d["sensor"]["film"]["rfilter"] = fd
    scene = mi.load_dict(d)
    osc, sensor = oracle_scene_from(O, scene)
    assert sensor.rfilter == tid and abs(sensor.rfilter_stddev - p0) < 1e-6 and (tid != 3 or abs(sensor.rfilter_param1 - p1) < 1e-6)
    L = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so")); L.hh_scene_create.restype = C.c_void_p
    L.hh_rfilter_eval.restype = C.c_float; L.hh_rfilter_eval.argtypes = [C.c_void_p, C.c_float]
    L.hh_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, O.c_f32p]
    for x in np.linspace(-radius - 0.3, radius + 0.3, 257):
        a = float(O.lib().orc_rfilter_eval2(tid, p0, p1, float(x))); b = float(L.hh_rfilter_eval(C.byref(sensor), float(x)))
        assert abs(a - b) <= 2e-7 + 1e-6 * abs(a), (name, x, a, b)
    desc = scene.desc(); err = C.create_string_buffer(256); h = C.c_void_p(L.hh_scene_create(C.byref(desc), err, 256)); assert h, err.value
    film = np.zeros((20, 20, 4), np.float32)
    assert L.hh_render(h, C.byref(sensor), 0, 2, 8, 5, 5, 0, 0, O.fp(film)) == 0
    ref, _ = osc.render_path(sensor, seed=2, spp=8, max_depth=5, raw=True)
    assert np.isfinite(film).all() and rel_l2(film, ref) < 1e-5
    assert film[..., 3].min() > 0                                  # every pixel received weight
