#!/usr/bin/env python3
"""Price of BLOCK-LEVEL REGROUPING -- the one traversal design left that changes who shares a wave step -- in the host harness (no GPU).
A block keeps R rays' traversal states in LDS; every round it deals them to its waves BY WHAT THEY DO NEXT, so that a wave issue runs ONE block of the loop (node visit /
triangle test / instance entry) for up to 64 rays that all need it, instead of all three blocks for whichever of its own 64 lanes need them.
    instructions per 64 rays = [node issues * (250 + S) + triangle issues * (60 + S) + instance issues * (100 + S) + rounds * waves_per_block * SORT] / (rays / 64)
with S = moving a ray's state between LDS and registers around an issue, SORT = a wave's share of the round's classification (ballots, prefix, barrier).
Compared with the same model of the per-lane kernel (tools/trace_stats.py: 250 / 60 / 100 per issued block + 30 per step), which lands within 15 % of SQ_INSTS_VALU.
Usage: HH_REGROUP=R is set by the script; python tools/regroup_model.py [instanced1m|flat1m] [res] [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(wl, res, spp, R):
    os.environ["HH_REGROUP"] = str(R)
    import mitsuba3_amd as mi
    mi.set_variant("hip_ad_rgb")
    d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=(wl == "flat1m"))
    scene = mi.load_dict(d)
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_scene_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    desc = scene.desc()
    h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256))
    assert h, err.value
    H.hh_set_order(2)
    out = np.zeros((8, 32), np.float64)
    sensor = scene.sensors()[0]
    H.hh_trace_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    assert H.hh_trace_stats(h, C.byref(sensor.har), 0, spp, 8, 5, 0, res * res * spp, 8, 0, 12, out.ctypes.data) == 0
    g = (C.c_double * 16)(); H.hh_regroup_stats(g)
    return out, np.asarray(list(g)).reshape(2, 8)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    R = int(os.environ.get("HH_REGROUP", "256"))
    out, g = run(wl, res, spp, R)
    measured = {"closest": 5519.0, "shadow": 6199.0}
    print("%s %dx%dx%d, blocks of %d rays (%d waves), refill at 20 %% idle" % (wl, res, res, spp, R, R // 64))
    for k, (kind, q) in enumerate((("closest", out[:, :16].sum(0)), ("shadow", out[:, 16:].sum(0)))):
        rays = q[0]; per = rays / 64.0
        cur = (250 * q[6] + 60 * q[7] + 100 * q[8] + 30 * q[5]) / per
        n_i, t_i, i_i, rounds, grays, ops = g[k][:6]
        gper = grays / 64.0
        print("%-7s per-lane kernel (model): %.0f instructions per 64 rays (measured %.0f); issues per 64 rays: node %.1f tri %.1f inst %.1f in %.1f steps" %
              (kind, cur, measured[kind], q[6] / per, q[7] / per, q[8] / per, q[5] / per))
        print("        regrouped: issues per 64 rays: node %.2f  tri %.2f  inst %.2f  (lane utilisation of an issue %.0f %%), rounds per block %.1f per 64 rays" %
              (n_i / gper, t_i / gper, i_i / gper, 100.0 * ops / (64.0 * (n_i + t_i + i_i)), rounds / gper))
        for S, SORT in ((0, 0), (40, 40), (60, 60), (80, 80)):
            cost = (n_i * (250 + S) + t_i * (60 + S) + i_i * (100 + S) + rounds * (R // 64) * SORT) / gper
            print("        state move S = %2d, sort = %2d per wave and round: %.0f instructions per 64 rays = %.2fx the per-lane model (%.2fx of the measured count scaled by the model's error)" %
                  (S, SORT, cost, cost / cur, cost / cur))
    print("LDS: %d rays x (~112 B state + 104 B stack) = %.0f KB per block -> %d blocks = %d waves per CU (the per-lane kernel: 6 blocks = 24 waves)" %
          (R, R * 216 / 1024.0, int(160 * 1024 // (R * 216)), int(160 * 1024 // (R * 216)) * (R // 64)))


if __name__ == "__main__":
    main()
