"""`har_render_backward` on a device (or allocator pool) that cannot hold the record tape: the library steps down -- record tape -> lane-indexed replay cache ->
smaller chunks -- instead of failing, and the gradients are the same (DESIGN "PRB record tape").  The test installs its own allocator through
`har_set_allocator` (the hook `mitsuba3_amd` uses for PyTorch's caching allocator): it serves blocks from torch's pool but refuses any request that would take the
library's outstanding bytes over a budget, which is what an `OutOfMemoryError` of the real pool looks like to the library (alloc_fn returns NULL)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import ctypes as C, gc, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["HAR_ROOT"])
import mitsuba3_amd as mi
from mitsuba3_amd import _capi

state = dict(live={}, total=0, peak=0, budget=None, refused=0)
def alloc(nbytes, user):
    if state["budget"] is not None and state["total"] + nbytes > state["budget"]:
        state["refused"] += 1
        return None
    p = torch.cuda.caching_allocator_alloc(int(nbytes), 0, torch.cuda.current_stream())
    state["live"][p] = nbytes; state["total"] += nbytes; state["peak"] = max(state["peak"], state["total"])
    return p
def free(p, user):
    if p:
        state["total"] -= state["live"].pop(p); torch.cuda.caching_allocator_delete(p)
ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)(alloc); FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)(free)

mi.set_variant("hip_ad_rgb")                       # HAR_TORCH_ALLOCATOR=0: the package installs nothing, the test's hook is the allocator
_capi.check(_capi.lib().har_set_allocator(C.cast(ALLOC, C.c_void_p), C.cast(FREE, C.c_void_p), None))
res, spp = 256, 64                                 # 2^22 lanes: room for two halvings above the 2^20-lane floor
d = mi.textured_cornell_box(res=res, tex_res=16, spp=spp)
grad_in = torch.from_numpy(np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)).cuda()
rel = lambda a, b: float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))

def run(budget):
    gc.collect()                                   # integrators of earlier runs give their workspaces back before the budget is set
    scene = mi.load_dict(d); integ = scene.integrator()
    base = state["total"]                          # scene arrays are allocated; the budget applies to what render_backward adds
    state["budget"] = None if budget is None else base + budget
    state["peak"] = state["total"]; state["refused"] = 0
    g = integ.render_backward(scene, None, grad_in, scene.sensors()[0], 5, spp)
    g2 = integ.render_backward(scene, None, grad_in, scene.sensors()[0], 5, spp)      # the second call starts from what the first settled on: no refusals
    torch.cuda.synchronize()
    out = ({k: v.clone() for k, v in g.items()}, {k: v.clone() for k, v in g2.items()}, state["peak"] - base, state["refused"])
    state["budget"] = None
    del g, g2, integ, scene
    return out

full, _, peak_full, refused = run(None)
assert refused == 0 and peak_full > 0
results = []
for frac in (0.6, 0.25):                           # first: the tape does not fit, the replay cache does; second: chunks have to shrink as well
    budget = int(peak_full * frac)
    g, g2, peak, refused = run(budget)
    assert refused >= 1, "the budget was meant to refuse the default workspace"
    assert peak <= budget
    for k in full:
        n = float(torch.linalg.norm(full[k]))
        if n > 0:
            assert rel(g[k], full[k]) < 2e-4, (frac, k, rel(g[k], full[k]))       # same lanes, same seeds: only the order of the atomic sums differs
            assert rel(g2[k], full[k]) < 2e-4, (frac, k)
    results.append((frac, peak, refused))
# a budget nothing fits in is an error with the allocator's reason, not a crash
gc.collect()
scene = mi.load_dict(d); integ = scene.integrator()
state["budget"] = state["total"] + (1 << 20)
try:
    integ.render_backward(scene, None, grad_in, scene.sensors()[0], 5, spp)
    raise SystemExit("expected an out-of-memory error")
except _capi.HarError as e:
    assert "no workspace fits" in str(e), str(e)
state["budget"] = None
print("FALLBACK_OK", peak_full, results)
'''


def test_render_backward_steps_down_when_the_tape_does_not_fit():
    env = dict(os.environ, HAR_ROOT=ROOT, HAR_TORCH_ALLOCATOR="0", HAR_VERBOSE="1")
    p = subprocess.run([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "FALLBACK_OK" in p.stdout, p.stdout[-4000:]
    assert "retrying with tape 0" in p.stdout, p.stdout[-4000:]                  # the record tape was given up first ...
    assert "lanes" in p.stdout
