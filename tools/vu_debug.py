import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import mitsuba3_amd as mi
from torch.profiler import profile, ProfilerActivity
mi.set_variant("hip_ad_rgb")
d = mi.cornell_box(); d["sensor"]["film"]["width"] = 128; d["sensor"]["film"]["height"] = 128
scene = mi.load_dict(d)
mi.render(scene, spp=16, seed=0); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(3):
        img = mi.render(scene, spp=16, seed=i); (img * 2).sum()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(len(evs), "device events")
for e in evs[:12]:
    print(e.name[:60], e.time_range.start, e.time_range.end)
