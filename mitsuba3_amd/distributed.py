"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI).  Lanes are independent, so rank r renders the contiguous lane
range [r*N/G, (r+1)*N/G) of the reference's own wavefront ordering -- i.e. a band of pixel
rows with ALL their samples -- KEEPING the global lane index for RNG seeding, so the union
of the bands is sample-identical to a single-GPU render.  Each rank splats into a private
full-size film; the only collective is ONE sum-reduce of the H x W x 4 film (and, for prb,
of the weight film and the gradient buffers).  The reference has no multi-GPU path
(SURVEY.md 2.3); this is new design.
"""
import torch
import torch.distributed as dist

from .core import develop_film, _torch


def lane_range(total_lanes, rank, world_size, granule=1):
    """Contiguous share of `total_lanes` for `rank`; boundaries are multiples of `granule`
    (use spp * width so that bands are whole pixel rows)."""
    units = total_lanes // granule
    lo = (units * rank) // world_size * granule
    hi = (units * (rank + 1)) // world_size * granule
    if rank == world_size - 1:
        hi = total_lanes
    return lo, hi


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def render_distributed(scene, integrator=None, sensor=0, seed=0, spp=0, develop=True, dst=0):
    """Integrator.render across all ranks; the developed image is valid on rank `dst`."""
    integrator = integrator or scene.integrator()
    s = scene.sensors()[sensor] if isinstance(sensor, int) else sensor
    if spp:
        s.sampler().set_sample_count(spp)
    spp = s.sampler().sample_count()
    w, h = s.film().crop_size()
    rank, world = _world()
    spp_pass, _ = integrator.pass_layout(s, spp)          # multi-pass jobs (> 2^32 - 1 samples): bands of the per-pass wavefront
    lanes = lane_range(w * h * spp_pass, rank, world, granule=w * spp_pass)
    film = integrator.render_film(scene, s, seed, spp, lanes=lanes)
    if world > 1:
        if dist.get_backend() == "gloo" and film.is_cuda:      # gloo has no device-tensor reduce (rehearsal runs only)
            dist.all_reduce(film, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)  # the single RCCL collective
    if not develop:
        return film
    return develop_film(film) if rank == dst or world == 1 else None


def render_backward_distributed(scene, grad_in, integrator=None, sensor=0, seed=0, spp=0):
    """RBIntegrator.render_backward across all ranks; gradients are all-reduced."""
    import ctypes as C
    from . import core
    from ._capi import lib, check
    integrator = integrator or scene.integrator()
    s = scene.sensors()[sensor] if isinstance(sensor, int) else sensor
    if spp:
        s.sampler().set_sample_count(spp)
    spp = s.sampler().sample_count()
    w, h = s.film().crop_size()
    rank, world = _world()
    lanes = lane_range(w * h * spp, rank, world, granule=w * spp)
    dev = core._device()
    wfilm = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
    sd = (s.sampler().m_base_seed + int(seed)) & 0xffffffff
    check(lib().har_render_weights(C.byref(s.har), sd, spp, lanes[0], lanes[1], core._ptr(wfilm), core._stream()))
    if world > 1:
        dist.all_reduce(wfilm, op=dist.ReduceOp.SUM)          # W[px] needs every rank's samples
    grads = integrator.render_backward(scene, None, grad_in, s, seed, spp, lanes=lanes, weight_film=wfilm)
    if world > 1:
        for g in grads.values():
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
    return grads
