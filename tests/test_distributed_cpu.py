"""Multi-GPU path on CPU: world_size-2 `gloo` run of the SAME host code (`lane_range` +
`render_distributed`: pixel-row bands with global lane seeding, private films, one sum-reduce)
with the CPU oracle standing in for the device renderer (there is no GPU in this container and
the product has no CPU fallback).  SURVEY.md 8(e)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HAR_ROOT"])
import mitsuba3_amd as mi
from oracle import oracle as O

class OracleIntegrator:
    """Test double with Integrator.render_film's signature; renders a lane band with the oracle."""
    def __init__(self, osc, sensor, samples_per_pass=None):
        self.osc, self.sensor, self.per_pass = osc, sensor, samples_per_pass
    def pass_layout(self, sensor, spp=0):
        return (self.per_pass, spp // self.per_pass) if self.per_pass else (spp, 1)
    def render_film(self, scene, sensor=0, seed=0, spp=0, lanes=None, film=None, alpha_film=None, film_window=None):
        if self.per_pass:
            raw, _ = self.osc.render_path_passes(self.sensor, seed=seed, spp=spp, spp_per_pass=self.per_pass, max_depth=8, lanes=lanes, raw=True)
        else:
            raw, _ = self.osc.render_path(self.sensor, seed=seed, spp=spp, max_depth=8, lanes=lanes, raw=True)
        if film_window is not None:
            # the band film of har_integrator_set_film_window: the rows of the window; NOTHING of this band's splats may fall outside it (Film.band_rows)
            lo, n = film_window
            outside = np.concatenate([raw[:lo].ravel(), raw[lo + n:].ravel()])
            assert not outside.any(), "a lane of rows %s splatted outside the film window %s" % (lanes, film_window)
            film[:n] += torch.from_numpy(raw[lo:lo + n])
            return film
        return torch.from_numpy(raw)

dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
mi.set_variant("hip_ad_rgb")
res, spp = 24, 4
d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
scene = mi.load_dict(d)                      # host-side scene only; no device handle is created
sd, osensor = O.cornell_box(res, res)
osc = O.OracleScene(sd)
film = mi.render_distributed(scene, integrator=OracleIntegrator(osc, osensor), seed=3, spp=spp, develop=False)
lo, hi = mi.lane_range(res * res * spp, rank, world, granule=res * spp)
assert lo % (res * spp) == 0 and (hi % (res * spp) == 0 or hi == res * res * spp)
if rank == 0:
    whole, _ = osc.render_path(osensor, seed=3, spp=spp, max_depth=8, raw=True)
    got = film.numpy()
    err = np.abs(got - whole).max() / np.abs(whole).max()
    assert err < 1e-6, err
    # the weight channel is exactly the sum of the bands' weights: every lane rendered once
    assert abs(got[..., 3].sum() - whole[..., 3].sum()) / whole[..., 3].sum() < 1e-6
# band films (the C5 layout: a rank owns its band + the filter's reach, one gather instead of the full-film reduce): same picture, for every filter width
for rf in ("gaussian", "box", "tent"):
    d2 = mi.cornell_box(); d2["sensor"]["film"]["width"] = res; d2["sensor"]["film"]["height"] = res; d2["sensor"]["film"]["rfilter"] = {"type": rf}
    scene2 = mi.load_dict(d2)
    sd2, osensor2 = O.scene_from_product(scene2)[0], O.scene_from_product(scene2)[1]
    film = mi.render_distributed(scene2, integrator=OracleIntegrator(sd2, osensor2), seed=3, spp=spp, develop=False, film_mode="band")
    if rank == 0:
        whole2, _ = sd2.render_path(osensor2, seed=3, spp=spp, max_depth=8, raw=True)
        assert film.shape == (res, res, 4) and np.abs(film.numpy() - whole2).max() / np.abs(whole2).max() < 1e-6, rf
    else:
        assert film is None
# multi-pass job (the C5 shape of SURVEY.md 8e at test size): ranks own bands of the PER-PASS wavefront and run every pass on them
film = mi.render_distributed(scene, integrator=OracleIntegrator(osc, osensor, samples_per_pass=2), seed=3, spp=8, develop=False)
if rank == 0:
    whole, _ = osc.render_path_passes(osensor, seed=3, spp=8, spp_per_pass=2, max_depth=8, raw=True)
    err2 = np.abs(film.numpy() - whole).max() / np.abs(whole).max()
    assert err2 < 1e-6, err2
# cost-balanced bands (BandBalancer): whatever boundaries the measured times produce, the union of the bands is the whole frame; after
# ADAPT_FRAMES frames the bands are frozen and identical on every rank
integ = OracleIntegrator(osc, osensor)
whole, _ = osc.render_path(osensor, seed=3, spp=spp, max_depth=8, raw=True)
for frame in range(5):
    film = mi.render_distributed(scene, integrator=integ, seed=3, spp=spp, develop=False)
    if rank == 0:
        e3 = np.abs(film.numpy() - whole).max() / np.abs(whole).max()
        assert e3 < 1e-6, (frame, e3)
(bal,) = integ._band_balancers.values()
assert bal.frames == 3 and not bal.adapting() and bal.bounds[0] == 0 and bal.bounds[-1] == res and all(b > a for a, b in zip(bal.bounds, bal.bounds[1:]))
mine = torch.tensor(bal.bounds, dtype=torch.int64); both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(both[0], b) for b in both)
if rank == 0:
    print("DIST_OK", err, err2)
dist.barrier()
dist.destroy_process_group()
'''


WORKER_PRB = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["HAR_ROOT"])
import mitsuba3_amd as mi
from oracle import oracle as O

class OracleIntegrator:
    """Test double with Integrator.render_film / render_weights / render_backward signatures; the oracle renders the rank's lane band."""
    def __init__(self, osc, sensor, keys):
        self.osc, self.sensor, self.keys = osc, sensor, keys
    def pass_layout(self, sensor, spp=0):
        return spp, 1
    def render_film(self, scene, sensor=0, seed=0, spp=0, lanes=None, film=None):
        raw, _ = self.osc.render_path(self.sensor, seed=seed, spp=spp, max_depth=6, lanes=lanes, raw=True)
        return torch.from_numpy(raw)
    def render_weights(self, scene, sensor=0, seed=0, spp=0, lanes=None):
        return torch.from_numpy(O.render_weights(self.sensor, seed, spp, lanes))
    def render_backward(self, scene, params, grad_in, sensor=0, seed=0, spp=0, lanes=None, weight_film=None):
        g_refl, g_tex, g_emit, _ = self.osc.render_prb_backward_lanes(self.sensor, np.asarray(grad_in), weight_film.numpy(), lanes, seed=seed, spp=spp, max_depth=6)
        return {"refl": torch.from_numpy(g_refl), "tex": torch.from_numpy(g_tex[0]), "emit": torch.from_numpy(g_emit)}

dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
mi.set_variant("hip_ad_rgb")
res, spp = 25, 4                                       # 25 rows: three ranks get ragged bands (8, 8, 9)
d = mi.textured_cornell_box(res=res, tex_res=8, spp=spp)
scene = mi.load_dict(d)                                # host-side scene only; no device handle is created
sd, osensor = O.cornell_box(res, res, white_texture=d["white"]["reflectance"]["data"])
osc = O.OracleScene(sd)
integ = OracleIntegrator(osc, osensor, None)
grad_in = np.random.default_rng(1).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
whole = osc.render_prb_backward_emitters(osensor, grad_in, seed=9, spp=spp, max_depth=6)
ref = {"refl": whole[0], "tex": whole[1][0], "emit": whole[2]}
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
for frame in range(4):                                 # bands move over the first three frames (BandBalancer), the sum must not
    grads = mi.render_backward_distributed(scene, grad_in, integ, seed=9, spp=spp)
    (bal,) = [b for k, b in integ._band_balancers.items() if k[0] == "prb"]
    assert bal.bounds[0] == 0 and bal.bounds[-1] == res and all(y > x for x, y in zip(bal.bounds, bal.bounds[1:]))
    for k in ref:                                      # all-reduced: every rank holds the whole-frame gradients
        e = rel(grads[k].numpy(), ref[k])
        assert e < 2e-5, (frame, k, e)
# the weight film every rank used is the whole frame's
y0, y1 = bal.band(rank)
wf = torch.from_numpy(O.render_weights(osensor, 9, spp, (y0 * res * spp, y1 * res * spp)))
dist.all_reduce(wf)
assert rel(wf.numpy(), O.render_weights(osensor, 9, spp)) < 1e-6
if rank == 0:
    print("DIST_PRB_OK", world, bal.bounds)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_lane_range_partitions_whole_rows():
    sys.path.insert(0, ROOT)
    from mitsuba3_amd.distributed import lane_range
    for (w, h, spp) in [(512, 512, 256), (33, 7, 5), (4096, 4096, 128)]:
        total = w * h * spp
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = lane_range(total, r, world, granule=w * spp)
                assert lo == prev and lo % (w * spp) == 0 and hi >= lo
                prev = hi
            assert prev == total


def test_band_balancer_equalises_cost():
    """BandBalancer.update on a synthetic cost-per-row profile (the bench scene's shape: cheap sky rows, expensive middle): one update brings the
    slowest rank from 1.2x to within 3 % of the mean, bands stay a partition of the rows, degenerate inputs are ignored"""
    sys.path.insert(0, ROOT)
    from mitsuba3_amd.distributed import BandBalancer
    rows = 512
    prof = np.interp(np.arange(rows), [0, 60, 200, 440, 512], [0.5, 0.8, 1.15, 1.25, 0.6])
    for world in (2, 3, 4, 8):
        b = BandBalancer(rows, world)
        cost = lambda: [prof[b.bounds[r]:b.bounds[r + 1]].sum() for r in range(world)]
        t0 = cost(); assert max(t0) / np.mean(t0) > 1.05
        b.update(t0); t1 = cost()
        assert max(t1) / np.mean(t1) < 1.03 and b.bounds[0] == 0 and b.bounds[-1] == rows and all(y > x for x, y in zip(b.bounds, b.bounds[1:]))
        b.update(t1); b.update(cost())
        assert not b.adapting() and max(cost()) / np.mean(cost()) < 1.02
    b = BandBalancer(4, 4); b.update([1.0, 5.0, 1.0, 1.0]); assert b.bounds == [0, 1, 2, 3, 4]          # one row per rank: nothing to move
    b = BandBalancer(64, 2); b.update([0.0, 1.0]); assert b.bounds == [0, 32, 64]                           # a missing measurement changes nothing
    b = BandBalancer(64, 2); b.update([1.0, 1e-9]); assert b.bounds[1] <= 63 and b.bounds[1] >= 1


def _run_ranks(worker, world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HAR_ROOT=ROOT,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, "-c", worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_world_size_2_gloo_band_union_equals_whole():
    assert "DIST_OK" in _run_ranks(WORKER, 2)[0]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_render_backward_distributed_equals_single_rank(world):
    """render_backward_distributed (weight-film all-reduce + gradient all-reduce) with 2 ranks and with 3 ranks over 25 rows (ragged bands):
    texel, constant-albedo and emitter gradients equal the single-rank oracle's whatever the band boundaries are"""
    assert "DIST_PRB_OK %d" % world in _run_ranks(WORKER_PRB, world)[0]


def test_c_abi_band_arithmetic_equals_the_python_balancer():
    """har_band_rebalance (the arithmetic har_multi_render re-cuts its bands with, csrc/har_multi.hip) == BandBalancer.update (the multi-process route), on random
    band layouts and times -- the two routes to N GPUs cut the same bands from the same measurements"""
    import ctypes as C
    import random
    import mitsuba3_amd as mi
    from mitsuba3_amd.distributed import BandBalancer
    L = mi.lib()
    rng = random.Random(7)
    for trial in range(400):
        n = rng.choice([2, 3, 4, 8]); rows = rng.choice([n, n + 1, 17, 64, 512, 4096])
        bal = BandBalancer(rows, n)
        if trial % 3:                       # start from uneven bands
            cuts = sorted(rng.sample(range(1, rows), n - 1)) if rows > n else list(range(1, n))
            bal.bounds = [0] + cuts + [rows]
        times = [rng.uniform(0.5, 20.0) * (1e-3 if trial % 2 else 1.0) for _ in range(n)]
        if trial % 41 == 0:
            times[rng.randrange(n)] = 0.0   # a band without a measurement: the bands stay
        before = list(bal.bounds)
        out = (C.c_uint32 * (n + 1))()
        assert L.har_band_rebalance(rows, n, (C.c_uint32 * (n + 1))(*before), (C.c_double * n)(*times), out) == 0
        bal.update(times)
        assert list(out) == bal.bounds, (trial, rows, n, before, times, list(out), bal.bounds)
        assert out[0] == 0 and out[n] == rows and all(b > a for a, b in zip(out, list(out)[1:]))
