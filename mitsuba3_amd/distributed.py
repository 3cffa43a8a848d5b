"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI).  Lanes are independent, so rank r renders the contiguous lane
range of a band of pixel rows with ALL their samples (equal bands at first, then bands of
equal measured cost, see BandBalancer) -- KEEPING the global lane index for RNG seeding, so the
union of the bands is sample-identical to a single-GPU render.  Each rank splats into a private
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


class BandBalancer:
    """Row bands of equal COST instead of equal height.  Equal bands leave the ranks unequal work (on the 1M-triangle bench scene the cost per
    pixel row varies by 2x from the top of the frame to the middle: 8 equal bands -> the slowest rank does 1.21x the mean, which caps strong
    scaling at 82 %).  Every rank times its own band (device events), the times are all-gathered, and the next frame's boundaries equalise the
    integral of the piecewise-constant cost-per-row estimate.  After `ADAPT_FRAMES` frames the bands are frozen, so a steady render loop pays
    neither the extra synchronisation nor the all-gather.  Bands stay whole pixel rows with global lane indices: the union of the bands is still
    sample-identical to a single-GPU render whatever the boundaries are."""

    ADAPT_FRAMES = 3

    def __init__(self, rows, world):
        self.rows, self.world = rows, world
        self.bounds = [rows * r // world for r in range(world + 1)]
        self.frames = 0

    def adapting(self):
        return self.world > 1 and self.frames < self.ADAPT_FRAMES

    def band(self, rank):
        return self.bounds[rank], self.bounds[rank + 1]

    def update(self, times):
        """times[r] = seconds rank r spent on its current band (identical list on every rank)"""
        self.frames += 1
        b, n = self.bounds, self.world
        if any(not (t > 0.0) for t in times) or self.rows < n:
            return
        dens = [times[r] / max(b[r + 1] - b[r], 1) for r in range(n)]            # cost per row, piecewise constant
        total = sum(times)
        new, r, acc = [0], 0, 0.0
        for k in range(1, n):
            target = total * k / n
            while r < n - 1 and acc + times[r] < target:
                acc += times[r]; r += 1
            y = b[r] + (target - acc) / dens[r] if dens[r] > 0 else b[r + 1]
            y = int(round(y))
            new.append(min(max(y, new[-1] + 1), self.rows - (n - k)))                # every rank keeps at least one row
        new.append(self.rows)
        self.bounds = new


def _balancer(integrator, key, rows, world):
    table = integrator.__dict__.setdefault("_band_balancers", {})
    bal = table.get(key)
    if bal is None or bal.rows != rows or bal.world != world:
        bal = table[key] = BandBalancer(rows, world)
    return bal


class _Timer:
    """device time of the enclosed launches (HIP events on the current stream), or wall time for a host renderer (the CPU test double)"""

    def __init__(self, enabled):
        self.enabled = enabled; self.cuda = enabled and torch.cuda.is_available()

    def __enter__(self):
        if self.cuda:
            self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True); self.e0.record()
        elif self.enabled:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.cuda:
            self.e1.record()
        elif self.enabled:
            import time
            self.dt = time.perf_counter() - self.t0

    def seconds(self):
        if self.cuda:
            self.e1.synchronize()
            return self.e0.elapsed_time(self.e1) * 1e-3
        return self.dt


def _share_times(bal, timer, like):
    """all-gather of the per-rank band times, then the new boundaries (same arithmetic on every rank)"""
    t = torch.tensor([timer.seconds()], dtype=torch.float64, device=like.device if dist.get_backend() != "gloo" else "cpu")
    out = [torch.zeros_like(t) for _ in range(bal.world)]
    dist.all_gather(out, t)
    bal.update([float(x.item()) for x in out])


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


BAND_FILM_MIN_BYTES = 32 << 20      # films at least this large are gathered as bands (render_distributed, film_mode=None)


def _gather_bands(band, rows, h, w, dst, rank, world):
    """`band` = this rank's rows [rows[rank][0], rows[rank][1]) of the film, padded to the common height: ONE gather to `dst`, which adds the bands (their halos overlap)
    into the H x W x 4 film.  An eighth of the reduce's bytes per rank at eight ranks, and no rank but `dst` ever holds a full film."""
    staged = band
    if dist.get_backend() == "gloo" and band.is_cuda:          # rehearsal runs (ranks share a GPU): gloo moves host tensors
        staged = band.cpu()
    got = [torch.empty_like(staged) for _ in range(world)] if rank == dst else None
    dist.gather(staged, got, dst=dst)
    if rank != dst:
        return None
    film = torch.zeros((h, w, 4), dtype=band.dtype, device=band.device)
    for r, (lo, hi) in enumerate(rows):
        if hi > lo:
            film[lo:hi] += got[r][:hi - lo].to(band.device)
    return film


def render_distributed(scene, integrator=None, sensor=0, seed=0, spp=0, develop=True, dst=0, film_mode=None):
    """Integrator.render across all ranks; the developed image is valid on rank `dst`.
    film_mode "full": every rank splats into a private full-size film, ONE sum-reduce (the default for ordinary films: a 512^2 film is 4 MiB, the reduce is latency);
    "band": every rank owns the rows its band can reach (band + filter reach + sample border; har_integrator_set_film_window), ONE gather, `dst` adds the bands --
    the default from BAND_FILM_MIN_BYTES up (BASELINE config 5: 256 MiB film -> 32 MiB + halo per rank at eight ranks)."""
    integrator = integrator or scene.integrator()
    s = scene.sensors()[sensor] if isinstance(sensor, int) else sensor
    if spp:
        s.sampler().set_sample_count(spp)
    spp = s.sampler().sample_count()
    w, h = s.film().crop_size()
    gw, gh = s.film().sample_grid()                       # rows of the SAMPLE grid are dealt to the ranks (= the crop unless Film::sample_border)
    rank, world = _world()
    spp_pass, _ = integrator.pass_layout(s, spp)          # multi-pass jobs (> 2^32 - 1 samples): bands of the per-pass wavefront
    bal = _balancer(integrator, ("path", id(scene), gw, spp_pass), gh, world)
    y0, y1 = bal.band(rank)
    lanes = (y0 * gw * spp_pass, y1 * gw * spp_pass)
    adapt = bal.adapting()
    if film_mode is None:
        film_mode = "band" if world > 1 and h * w * 16 >= BAND_FILM_MIN_BYTES and hasattr(s.film(), "band_rows") else "full"
    if film_mode == "band" and world > 1:
        from . import core
        rows = [s.film().band_rows(bal.bounds[r], bal.bounds[r + 1]) for r in range(world)]          # identical on every rank
        lo, hi = rows[rank]
        rows_max = max(1, max(b - a for a, b in rows))
        dev = core._device() if torch.cuda.is_available() else torch.device("cpu")
        band = torch.zeros((rows_max, w, 4), dtype=torch.float32, device=dev)
        alpha = torch.zeros_like(band) if getattr(s.film(), "alpha", False) and develop else None
        with _Timer(adapt) as timer:
            if y1 > y0:
                band = integrator.render_film(scene, s, seed, spp, lanes=lanes, film=band, alpha_film=alpha, film_window=(lo, max(hi - lo, 1)))
        if adapt:
            _share_times(bal, timer, band)
        film = _gather_bands(band, rows, h, w, dst, rank, world)
        alpha = _gather_bands(alpha, rows, h, w, dst, rank, world) if alpha is not None else None
        if not develop:
            return film
        return develop_film(film, alpha, getattr(s.film(), "colour", 0)) if rank == dst else None
    alpha = None
    if getattr(s.film(), "alpha", False) and develop:             # `rgba` films: a second accumulator (w * alpha), reduced like the first
        from . import core
        alpha = torch.zeros((h, w, 4), dtype=torch.float32, device=core._device())
    with _Timer(adapt) as timer:
        film = integrator.render_film(scene, s, seed, spp, lanes=lanes) if alpha is None else integrator.render_film(scene, s, seed, spp, lanes=lanes, alpha_film=alpha)
    if adapt:
        _share_times(bal, timer, film)
    if world > 1:
        for buf in ((film,) if alpha is None else (film, alpha)):
            if dist.get_backend() == "gloo" and buf.is_cuda:      # gloo has no device-tensor reduce (rehearsal runs only)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            else:
                dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)  # the single RCCL collective (two for `rgba` films)
    if not develop:
        return film
    return develop_film(film, alpha, getattr(s.film(), "colour", 0)) if rank == dst or world == 1 else None


def render_backward_distributed(scene, grad_in, integrator=None, sensor=0, seed=0, spp=0):
    """RBIntegrator.render_backward across all ranks (common.py:625-783): every rank splats the filter weights of its lane band, ONE all-reduce
    makes W[px] complete everywhere (the adjoint of develop needs every rank's samples), every rank replays its band, and the gradient
    buffers are all-reduced.  Returns {key: gradient tensor}, identical on every rank."""
    integrator = integrator or scene.integrator()
    s = scene.sensors()[sensor] if isinstance(sensor, int) else sensor
    if spp:
        s.sampler().set_sample_count(spp)
    spp = s.sampler().sample_count()
    gw, gh = s.film().sample_grid()
    rank, world = _world()
    bal = _balancer(integrator, ("prb", id(scene), gw, spp), gh, world)
    y0, y1 = bal.band(rank)
    lanes = (y0 * gw * spp, y1 * gw * spp)
    adapt = bal.adapting()
    wfilm = integrator.render_weights(scene, s, seed, spp, lanes=lanes)
    if world > 1:
        dist.all_reduce(wfilm, op=dist.ReduceOp.SUM)          # W[px] needs every rank's samples
    with _Timer(adapt) as timer:
        grads = integrator.render_backward(scene, None, grad_in, s, seed, spp, lanes=lanes, weight_film=wfilm)
    if adapt:
        _share_times(bal, timer, wfilm)
    if world > 1 and grads:            # (an empty dict -- no differentiable key -- needs no collective; every rank sees the same key set)
        # ONE collective for all gradient buffers (texels, constant albedos, emitter radiance, ...): xGMI rings are latency bound for the small ones
        keys = list(grads)
        flat = torch.cat([grads[k].reshape(-1).to(torch.float32) for k in keys])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for k in keys:
            n = grads[k].numel()
            grads[k] = flat[off:off + n].reshape(grads[k].shape).to(grads[k].dtype); off += n
    return grads
