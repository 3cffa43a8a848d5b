#!/usr/bin/env python3
"""Model of a wave-shared ("packet") BVH descent for coherent launches (host test harness; no GPU): the 64 rays of a wave -- at >= 64 spp the samples of ONE pixel --
walk the BVH together with one conservative interval test per child box, exact per-lane triangle tests at the leaves.  Prints, per bounce and query kind, the blocks a
packet wave would issue next to the per-ray loop's counts for the same rays, and a VALU-instruction estimate for both.
Usage: python tools/packet_stats.py [instanced1m|flat1m|cornell] [res] [spp] [bounces] [first_row] [rows]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3_amd as mi                                     # noqa: E402

# instruction prices (VALU wave instructions): per-lane kernel measured 6 233 per 64 closest-hit rays at 10.1 node visits per ray (profiles/r03_sq_instanced1m.json);
# packet blocks: node visit = scalar node fetch + 8-lane interval test + mask assembly, leaf = one exact triangle test for all 64 lanes, instance entry = per-lane ray
# transform + wave min / max of origin and reciprocal direction (12 DPP reductions)
PK_NODE, PK_TRI, PK_INST, PK_FIXED = 70.0, 75.0, 260.0, 150.0
LANE_PER_NODE_VISIT = 6233.0 / 10.1


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    nb = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    row0 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    rows = int(sys.argv[6]) if len(sys.argv) > 6 else res
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
    out = np.zeros((nb, 2, 12), np.float64)
    cap = res * res * spp // 64 + 8
    pp = np.zeros((cap, 8), np.float64)
    sensor = scene.sensors()[0]
    H.hh_packet_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
    rc = H.hh_packet_stats(h, C.byref(sensor.har), 0, spp, 8, 5, row0 * res * spp, (row0 + rows) * res * spp, nb, out.ctypes.data, pp.ctypes.data, cap)
    assert rc == 0
    print("%s %dx%dx%d rows [%d, %d)  (packet = 64 consecutive rays of the launch)" % (wl, res, res, spp, row0, row0 + rows))
    print("%-2s %-8s %9s %6s | packet: %7s %7s %7s | per ray: %6s %6s %6s | est. VALU / 64 rays: %8s %8s %6s" %
          ("b", "kind", "rays", "mism", "nodes", "leaves", "insts", "nodes", "tris", "insts", "packet", "per-lane", "ratio"))
    for b in range(nb):
        for k, kind in ((0, "closest"), (1, "shadow")):
            q = out[b, k]
            if q[0] == 0:
                continue
            pk = (PK_NODE * q[2] + PK_TRI * q[3] + PK_INST * q[4]) / q[0] + PK_FIXED
            lane = LANE_PER_NODE_VISIT * q[6] / q[1]
            print("%-2d %-8s %9d %6d | %15.1f %7.1f %7.2f | %15.2f %6.2f %6.2f | %28.0f %8.0f %6.2f" %
                  (b, kind, q[1], q[5], q[2] / q[0], q[3] / q[0], q[4] / q[0], q[6] / q[1], q[7] / q[1], q[8] / q[1], pk, lane, lane / pk))
    # per-packet distribution at bounce 0: how many packets are worth the packet path, and what a per-packet choice would give
    for k, kind in ((0, "closest"), (4, "shadow")):
        n = int(out[0, 0 if k == 0 else 1, 0])
        q = pp[:n, k:k + 4]
        pk = PK_NODE * q[:, 0] + PK_TRI * q[:, 1] + PK_INST * q[:, 2] + PK_FIXED
        lane = LANE_PER_NODE_VISIT * q[:, 3] / 64.0
        best = np.minimum(pk, lane)
        print("bounce 0 %-8s per-packet choice: %5.1f %% of the packets cheaper as a packet; all-packet %.0f, all-per-lane %.0f, best-of-both %.0f VALU per 64 rays (x%.2f)" %
              (kind, 100.0 * (pk < lane).mean(), pk.mean(), lane.mean(), best.mean(), lane.mean() / best.mean()))
        print("    packet cost percentiles (VALU per packet) 10/50/90/99: %s   per-lane %s" % (np.percentile(pk, [10, 50, 90, 99]).round(0), np.percentile(lane, [10, 50, 90, 99]).round(0)))


if __name__ == "__main__":
    main()
