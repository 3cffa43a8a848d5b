#!/usr/bin/env python3
"""Traversal statistics of a scene on the CPU (host test harness; no GPU): per-ray node visits /
triangle tests / instance entries per bounce, and a lock-step 64-lane SIMT model of the static
traversal kernel (how many node-visit / triangle-test blocks a wave executes vs. the useful work).
Usage: python tools/trace_stats.py [instanced1m|flat1m|cornell] [res] [spp] [rows]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3_amd as mi                                     # noqa: E402
from mitsuba3_amd import _capi                                # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    policy = int(sys.argv[4]) if len(sys.argv) > 4 else -1      # -1 reference loop, 0 / 1 Traversal<POLICY>
    refill = int(sys.argv[5]) if len(sys.argv) > 5 else 0       # 0 static waves, R = persistent wave refilled at >= R idle lanes
    mi.set_variant("hip_ad_rgb")
    if wl == "cornell":
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    else:
        d = mi.instanced_spheres_scene(width=res, height=res, spp=spp, flatten=(wl == "flat1m"))
    scene = mi.load_dict(d)
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_scene_create.restype = C.c_void_p
    err = C.create_string_buffer(256)
    desc = scene.desc()
    h = C.c_void_p(H.hh_scene_create(C.byref(desc), err, 256))
    assert h, err.value
    nb = 8
    H.hh_set_order(int(sys.argv[6]) if len(sys.argv) > 6 else 2)
    out = np.zeros((nb, 32), np.float64)
    sensor = scene.sensors()[0]
    H.hh_trace_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    rc = H.hh_trace_stats(h, C.byref(sensor.har), 0, spp, 8, 5, 0, res * res * spp, nb, policy, refill, out.ctypes.data)
    assert rc == 0
    print("%s %dx%dx%d policy %d refill %d mismatches %d" % (wl, res, res, spp, policy, refill, int(out[:, 10].sum() + out[:, 26].sum())))
    print("%-3s %-8s %9s | per ray: %6s %6s %6s %6s | per wave: %7s %8s %8s %8s | SIMT efficiency node/tri" % ("b", "kind", "rays", "iters", "nodes", "tris", "insts", "steps", "nodeblk", "triblk", "instblk"))
    tot = np.zeros(32)
    for b in range(nb):
        for kind, q in (("closest", out[b, :16]), ("shadow", out[b, 16:])):
            if q[0] == 0:
                continue
            r, w = q[0], q[9]
            print("%-3d %-8s %9d | %15.2f %6.2f %6.2f %6.2f | %17.1f %8.1f %8.1f %8.1f | %5.2f %5.2f" %
                  (b, kind, r, q[1] / r, q[2] / r, q[3] / r, q[4] / r, q[5] / w, q[6] / w, q[7] / w, q[8] / w,
                   q[2] / (64 * q[6]) if q[6] else 0, q[3] / (64 * q[7]) if q[7] else 0))
    print("max traversal stack entries used: %d (scene stack_need %d)" % (H.hh_max_sp(), scene.accel_info_host()["depth"] if hasattr(scene, "accel_info_host") else -1))
    tp = (C.c_double * 24)(); H.hh_top_phase_stats(tp)
    for q, kind in ((0, "closest-hit"), (1, "any-hit")):
        r, top_n, top_t, tlas_n, inst_n, inst_t, ent, empty, stale, fent, fnodes, _ = tp[12 * q:12 * q + 12]
        if r:
            print("reference loop (top-level first), %s queries: top-level BLAS %.2f node visits + %.2f triangle tests per ray, TLAS %.2f node visits, %.2f instance entries per ray "
                  "with %.2f node visits + %.2f triangle tests per ENTRY; %.2f node visits per ray (%.0f %%) hit none of the node's children, %.2f of them lie beyond the current tmax"
                  % (kind, top_n / r, top_t / r, tlas_n / r, ent / r, inst_n / max(ent, 1), inst_t / max(ent, 1), empty / r, 100 * empty / max(top_n + tlas_n + inst_n, 1), stale / r))
            if q == 0:
                print("    instance entries that gave the ray no closer hit: %.0f %% of the entries, %.2f node visits each (%.2f per ray)" % (100 * fent / max(ent, 1), fnodes / max(fent, 1), fnodes / r))
    tc = out[:, :16].sum(0); ts = out[:, 16:].sum(0)
    for kind, q in (("closest", tc), ("shadow", ts)):
        r, w = q[0], q[9]
        print("all %-8s %9d | %15.2f %6.2f %6.2f %6.2f | %17.1f %8.1f %8.1f %8.1f | %5.2f %5.2f" %
              (kind, r, q[1] / r, q[2] / r, q[3] / r, q[4] / r, q[5] / w, q[6] / w, q[7] / w, q[8] / w, q[2] / (64 * q[6]), q[3] / (64 * q[7])))
        # VALU cost model (instructions issued per 64 rays): node block 250, triangle block 60, instance block 100, loop overhead 30 per step
        if kind == "shadow" and q[11]:
            print("    occluded shadow rays: %.1f %%, node visits per occluded ray %.2f, per unoccluded ray %.2f" % (100 * q[11] / r, q[12] / q[11], (q[2] - q[12]) / max(r - q[11], 1)))
            print("    occluder-cache what-if (per pixel and bounce, previous occluded shadow ray): same instance / mesh %.1f %% of the occluded rays, same triangle %.1f %%" % (100 * q[13] / q[11], 100 * q[14] / q[11]))
        cost = (250 * q[6] + 60 * q[7] + 100 * q[8] + 30 * q[5]) / (r / 64)
        print("    modelled VALU instructions per 64 rays: %.0f  (blocks per 64 rays: steps %.1f node %.1f tri %.1f inst %.1f)" %
              (cost, q[5] / (r / 64), q[6] / (r / 64), q[7] / (r / 64), q[8] / (r / 64)))


if __name__ == "__main__":
    main()
