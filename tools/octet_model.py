#!/usr/bin/env python3
"""Price of a CHILD-PARALLEL traversal (8 lanes per ray, lane s owns child slot s of the node its octet visits; round-5 verdict, item 2) against the per-lane kernel, on
the CPU (host harness, no GPU).  The model is a LOWER bound for the octet design: perfect refill (no octet ever idles), leaf tests fused into the node step for free
(every hit leaf slot is tested by its own lane in the step that found it), one extra step per instance entry (the ray of all eight lanes changes its space).
    wave steps per 64 rays  =  64 / 8 * (node visits + instance entries per ray)
    VALU per 64 rays        =  wave steps * (instructions of one step)
against the MEASURED 5 519 (closest hit) / 6 199 (shadow) wave-VALU instructions per 64 rays of the per-lane kernels (profiles/r05_sq_instanced1m.json).
Usage: python tools/octet_model.py [instanced1m|flat1m] [res] [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mitsuba3_amd as mi                                     # noqa: E402

# instructions of one octet step, block by block (counted from the per-lane kernel's ISA blocks, profiles/r04_isa_blocks.txt, re-shaped for one child per lane)
STEP = [
    ("pop the octet's stack entry, child index, node address", 6),
    ("80-byte node fetch by the octet + this lane's six plane bytes out of it", 8),
    ("frame: three exponents -> scales, a = s * idir, b = (p - o) * idir (per RAY, redone by all 8 lanes)", 12),
    ("near / far plane by the ray's sign (6 selects)", 6),
    ("slab test of ONE child: 6 cvt + 6 fma + 4 min/max + the padded compare", 18),
    ("ballot, this octet's byte of it, inner / leaf split, front-to-back order, push", 13),
    ("leaf lanes: 48-byte record + Moeller-Trumbore (issued whenever any of 64 lanes holds a hit leaf slot: ~every step)", 48),
    ("closest hit of the octet: DPP min over 8 lanes, tie rule, winner's record to all 8 lanes, tmax", 22),
    ("refill test + amortised refill / commit", 10),
]
INST_ENTRY = 100     # the per-lane kernel's instance block: ray transform, three reciprocals, two pushes (r04_isa_blocks)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "instanced1m"
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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
    base = sum(c for _, c in STEP)
    print("%s %dx%dx%d -- one octet step, by block:" % (wl, res, res, spp))
    for name, c in STEP:
        print("    %3d  %s" % (c, name))
    print("    %3d  = a step without an instance entry; + %d in the steps where one of the 8 octets enters an instance" % (base, INST_ENTRY))
    measured = {"closest": 5519.0, "shadow": 6199.0}
    for kind, q in (("closest", out[1:, :16].sum(0)), ("shadow", out[:, 16:].sum(0))):      # closest: bounces >= 1 (bounce 0 stays with the packet kernel)
        r = q[0]
        nodes, tris, insts = q[2] / r, q[3] / r, q[4] / r
        steps_ray = nodes + insts
        wave_steps = 64.0 / 8.0 * steps_ray
        p_inst = 1.0 - (1.0 - insts / steps_ray) ** 8
        per_step = base + INST_ENTRY * p_inst
        valu = wave_steps * per_step
        bare = wave_steps * (18 + 48)                 # nothing but the slab test and the triangle test
        cur_steps = q[5] / (r / 64)
        print("%-7s per ray: %.2f node visits, %.2f triangle tests, %.2f instance entries -> octet: %.1f wave steps per 64 rays (per-lane kernel: %.1f), "
              "P(instance entry in a step) %.2f, %.0f instructions per step" % (kind, nodes, tris, insts, wave_steps, cur_steps, p_inst, per_step))
        print("        octet VALU per 64 rays >= %.0f against %.0f measured for the per-lane kernel: %.2fx MORE (speed-up %.2fx; the verdict's bar is 1.20x)"
              % (valu, measured[kind], valu / measured[kind], measured[kind] / valu))
        print("        even a step of NOTHING but one slab test + one triangle test (66 instructions): %.0f = %.2fx the per-lane kernel; break-even step = %.0f instructions, 1.2x needs %.0f"
              % (bare, bare / measured[kind], measured[kind] / wave_steps, measured[kind] / wave_steps / 1.2))


if __name__ == "__main__":
    main()
