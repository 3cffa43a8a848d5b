"""Multi-GPU path with DEVICE tensors: two ranks run `render_distributed` / `render_backward_distributed` on the HIP renderer --
band lanes with global seeding, private device films, the device-side timers of the BandBalancer, the film reduce and the flat gradient all-reduce --
and the result must equal one rank rendering the whole frame.

* `rccl`: one rank per GPU over RCCL (backend "nccl"); skipped when the box has fewer than two GPUs (gpurun boxes have one; the driver's 8-GPU node has eight).
* `gloo-shared`: both ranks on cuda:0 with the gloo backend (device tensors staged by gloo).  It runs on a 1-GPU box and covers everything but RCCL itself:
  the device-tensor code paths of distributed.py (`_Timer` events, `_share_times`, `torch.cat` of device gradients) are the same."""
import os
import subprocess
import sys

import pytest

from tests.test_distributed_cpu import _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HAR_ROOT"])
import mitsuba3_amd as mi

backend = os.environ["HAR_BACKEND"]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank if backend == "nccl" else 0)
dist.init_process_group(backend=backend, rank=rank, world_size=world)
mi.set_variant("hip_ad_rgb")
res, spp = 50, 16                                     # 50 rows: bands of 25 that the balancer then moves
d = mi.textured_cornell_box(res=res, tex_res=8, spp=spp)
scene = mi.load_dict(d)
integ = scene.integrator()
rel = lambda a, b: float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))

# forward: union of the ranks' bands == the whole frame rendered by this rank alone (same lanes, same seeds; only the atomic order differs)
whole = integ.render_film(scene, scene.sensors()[0], 3, spp)
assert whole.is_cuda
for frame in range(5):                                # BandBalancer adapts over three frames, then freezes
    film = mi.render_distributed(scene, integ, seed=3, spp=spp, develop=False)
    assert film.is_cuda
    if rank == 0 or backend == "gloo":                # gloo rehearsal all-reduces, RCCL reduces to rank 0
        e = rel(film, whole)
        assert e < 1e-5, (frame, e)
        assert abs(float(film[..., 3].sum()) - float(whole[..., 3].sum())) < 1e-4 * float(whole[..., 3].sum())
(bal,) = [b for k, b in integ._band_balancers.items() if k[0] == "path"]
assert not bal.adapting() and bal.bounds[0] == 0 and bal.bounds[-1] == res and all(y > x for x, y in zip(bal.bounds, bal.bounds[1:]))
img = mi.render_distributed(scene, integ, seed=3, spp=spp)
assert (img is not None) == (rank == 0)
# band films (the layout of BASELINE config 5): every rank owns its band + the filter's reach (har_integrator_set_film_window), ONE gather, rank 0 adds the bands
bfilm = mi.render_distributed(scene, integ, seed=3, spp=spp, develop=False, film_mode="band")
assert (bfilm is not None) == (rank == 0)
if rank == 0:
    assert bfilm.is_cuda and bfilm.shape == whole.shape and rel(bfilm, whole) < 1e-5
# a window that cannot hold the band's splats is refused, not overrun
lo, hi = scene.sensors()[0].film().band_rows(10, 20)
small = torch.zeros((hi - lo - 1, res, 4), device="cuda")
try:
    integ.render_film(scene, scene.sensors()[0], 3, spp, lanes=(10 * res * spp, 20 * res * spp), film=small, film_window=(lo + 1, hi - lo - 1))
    raise SystemExit("a film window smaller than the band's reach was accepted")
except mi.HarError as e:
    assert "window" in str(e)
exact = torch.zeros((hi - lo, res, 4), device="cuda")
integ.render_film(scene, scene.sensors()[0], 3, spp, lanes=(10 * res * spp, 20 * res * spp), film=exact, film_window=(lo, hi - lo))
part = integ.render_film(scene, scene.sensors()[0], 3, spp, lanes=(10 * res * spp, 20 * res * spp))
assert rel(exact, part[lo:hi]) < 1e-6 and float(part[:lo].abs().sum()) == 0 and float(part[hi:].abs().sum()) == 0

# adjoint: weight-film all-reduce + ONE flat all-reduce of every gradient buffer; every rank ends up with the whole frame's gradients
grad_in = torch.from_numpy(np.random.default_rng(1).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)).cuda()
single = integ.render_backward(scene, None, grad_in, scene.sensors()[0], 9, spp)
assert single, "the scene has differentiable parameters"
for frame in range(4):
    grads = mi.render_backward_distributed(scene, grad_in, integ, seed=9, spp=spp)
    assert set(grads) == set(single)
    for k in single:
        assert grads[k].is_cuda and grads[k].shape == single[k].shape
        if float(torch.linalg.norm(single[k])) > 0:
            e = rel(grads[k], single[k])
            assert e < 1e-4, (frame, k, e)
torch.cuda.synchronize()
dist.barrier()
if rank == 0:
    print("DIST_GPU_OK", backend, bal.bounds, sorted(single))
dist.destroy_process_group()
'''


def _run(backend, world=2):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HAR_ROOT=ROOT, HAR_BACKEND=backend,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_world_2_rccl_device_tensors_equal_single_rank():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL)")
    assert "DIST_GPU_OK nccl" in _run("nccl")[0]


def test_world_2_shared_gpu_device_tensors_equal_single_rank():
    assert "DIST_GPU_OK gloo" in _run("gloo")[0]
