#!/usr/bin/env python3
"""One host thread, N GPUs: the single-call multi-GPU entry of the C ABI (har_multi_render / har_multi_render_backward) through mi.DeviceGroup.
A replica of the scene lives on every device; a frame is dealt to the devices as bands of pixel rows (global lane indices: the union of the bands draws exactly
the samples of a single-GPU render), the films meet on devices[0] in ONE ncclReduce.  With fewer GPUs than replicas the same device may be named twice (the
collective is then a peer copy + add): that is how the band machinery is exercised on a one-GPU box.
Usage: python examples/render_device_group.py [device indices, e.g. 0 1 2 3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mitsuba3_amd as mi                                     # noqa: E402


def main():
    devices = [int(a) for a in sys.argv[1:]] or [0, 0]
    mi.set_variant("hip_ad_rgb")
    d = mi.textured_cornell_box(res=256, tex_res=64, spp=64)
    d["integrator"]["emitter_gradients"] = True
    scene = mi.load_dict(d)
    group = mi.DeviceGroup(scene, devices=devices)
    for frame in range(4):                                   # the bands are re-cut from the measured device times of the first frames
        img = group.render(spp=64, seed=frame)
    print("image %s on %s, mean %.4f" % (tuple(img.shape), img.device, float(img.mean())))
    print("bands:", group.info())
    grads = group.render_backward(2.0 * img / img.numel(), seed=17, spp=64)      # d mean(img^2) / d parameters
    for k, g in grads.items():
        print("  d loss / d %-32s |g|max = %.3e" % (k, float(g.abs().max())))
    single = mi.render(scene, spp=64, seed=3)
    print("group vs single-device render (same seed): rel L2 = %.2e" % float((group.render(spp=64, seed=3) - single).norm() / single.norm()))


if __name__ == "__main__":
    main()
