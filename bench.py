#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hip_ad_rgb hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete forward `path` render of the workload (all W*H*spp lanes:
raygen -> [trace, shade, resolve]* -> splat -> film reduce -> develop), scene and BVH
already resident in HBM.  value = W*H*spp / seconds / 1e6 (Mpaths/s), whole job over all
ranks (weak scaling is NOT used: the image is fixed and sharded by pixel rows => "strong";
the row bands are rebalanced by measured per-rank time over the first three frames).
After the timed forward steps the PRB adjoint (RBIntegrator.render_backward equivalent:
weight pass + primal pass + adjoint replay) is timed the same way and reported in
`prb_adjoint`.  rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="instanced1m", choices=["instanced1m", "flat1m", "cornell", "materials1m"])
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0, help="wavefront chunk size in lanes (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prb", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def build_scene(mi, args, integrator_type):
    if args.workload == "cornell":
        d = mi.cornell_box()
        d["sensor"]["film"]["width"] = args.res; d["sensor"]["film"]["height"] = args.res
        d["sensor"]["sampler"]["sample_count"] = args.spp
    else:
        d = mi.instanced_spheres_scene(width=args.res, height=args.res, spp=args.spp, grid=10, n_u=100, n_v=50,
                                       flatten=(args.workload in ("flat1m", "materials1m")), max_depth=args.max_depth,
                                       materials=(args.workload == "materials1m"))
    d["integrator"] = {"type": integrator_type, "max_depth": args.max_depth, "rr_depth": 5, "chunk_lanes": args.chunk}
    if integrator_type == "prb":
        d["integrator"]["emitter_gradients"] = False      # north_star: gradients w.r.t. BSDF / texture parameters
    return mi.load_dict(d)


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import mitsuba3_amd as mi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hip_ad_rgb path has no CPU fallback")
    # HAR_BENCH_SHARE_GPU=1 + HAR_BENCH_BACKEND=gloo: rehearsal of the N-rank code path on a box with fewer GPUs than ranks
    # (ranks share devices, the film reduce goes through gloo); never used for reported numbers
    device_index = local_rank % torch.cuda.device_count() if os.environ.get("HAR_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("HAR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    mi.set_variant("hip_ad_rgb")

    def sync_barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync_barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync_barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    n_paths = args.res * args.res * args.spp

    # ---------------- forward `path` ----------------
    scene = build_scene(mi, args, "path")
    integ = scene.integrator()
    accel = scene.accel_info()
    integ.set_profiling(True)          # HIP events on the launch stream, one per kernel launch

    def fwd():
        return mi.render_distributed(scene, integ, seed=0, spp=args.spp)

    dt = timed(fwd, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = n_paths / (dt / args.steps) / 1e6
    timing = integ.timing()            # of the LAST timed step on this rank
    stats = integ.stats()

    # roofline of the dominant kernel, algorithmic bytes per SURVEY.md 8(d):
    #   trace_closest: 56 B/ray (32 B ray in + 24 B hit out) + unique accel bytes once per launch
    #   shade: 72+24 B in, 72 B out (live) ; resolve: 48 B/item + 32 B result RMW
    kern_ms = {k: v[0] for k, v in timing.items()}
    dominant = max(("trace_closest", "shade", "resolve", "splat", "raygen"), key=lambda k: kern_ms[k])
    launches = max(timing[dominant][1], 1)
    if dominant == "trace_closest":
        alg_bytes = stats["closest_rays"] * 56 + accel["bytes"] * launches
    elif dominant == "resolve":
        alg_bytes = stats["shadow_rays"] * 33 + accel["bytes"] * launches
    elif dominant == "shade":
        alg_bytes = stats["vertices"] * (152 * 2 + 24 + 112)
    else:
        alg_bytes = stats["paths"] * (152 + 16)
    achieved = alg_bytes / 1e9 / (kern_ms[dominant] / 1e3) if kern_ms[dominant] > 0 else 0.0
    kbar = stats["vertices"] / max(stats["paths"], 1)
    b_path = 152 + kbar * 505 + 16
    # HBM traffic of the dominant kernel from the committed PMC passes (tools/gpu_profile.sh ... mem): bytes per launch
    # = (FETCH_SIZE x 2 [gfx950 correction, MI355X_MICROARCH.md "HBM"] + WRITE_SIZE) KiB x 1024 / launches; null if the
    # profile of this exact workload has not been collected
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % args.workload)
    if os.path.exists(tpath) and args.res == 512 and args.spp == 256 and world == 1:      # the PMC passes were collected at N = 1 (launch sizes differ otherwise)
        try:
            with open(tpath) as f:
                tj = json.load(f)
            krec = next(v for k, v in tj.items() if ("k_" + dominant) in k and "counters" in v)
            fetch = krec["counters"]["FETCH_SIZE"]; write = krec["counters"]["WRITE_SIZE"]
            traffic = round((2.0 * fetch["sum"] / fetch["dispatches"] + write["sum"] / write["dispatches"]) * 1024.0)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes_per_launch": round(alg_bytes / launches),
                "avg_launch_ms": round(kern_ms[dominant] / launches, 4), "launches": launches,
                "kernel_ms": {k: round(v, 3) for k, v in kern_ms.items()},
                "whole_path_model": {"K_bar": round(kbar, 3), "B_path": round(b_path, 1),
                                     "achieved_GBs": round(value * 1e6 * b_path / 1e9, 1),
                                     "frac_of_8TBs": round(value * 1e6 * b_path / 1e9 / HBM_PEAK_GBS, 5)}}

    # ---------------- PRB adjoint ----------------
    prb = None
    if not args.no_prb:
        scene_p = build_scene(mi, args, "prb")
        integ_p = scene_p.integrator()
        grad_in = torch.full((args.res, args.res, 3), 1.0 / (args.res * args.res * 3), device="cuda")

        def bwd():
            return mi.render_backward_distributed(scene_p, grad_in, integ_p, seed=1, spp=args.spp)

        p_steps = max(1, min(args.steps, 2))
        dtp = timed(bwd, p_steps, 1 if args.warmup else 0)
        prb = {"metric": "Mpaths/s prb adjoint (weight pass + primal + adjoint replay)",
               "value": round(n_paths / (dtp / p_steps) / 1e6, 2), "ms_per_step": round(dtp / p_steps * 1e3, 2), "steps": p_steps}

    # ---------------- CPU baseline (oracle, rank 0, N = 1 only) ----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        from tests.test_cpu_host import oracle_scene_from
        osc, sensor = oracle_scene_from(O, scene)
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        _, st = osc.render_path(sensor, seed=0, spp=1, max_depth=args.max_depth, threads=cores)
        probe = time.perf_counter() - t0
        spp_cpu = int(max(1, min(32, args.cpu_seconds / max(probe, 1e-3))))
        t0 = time.perf_counter()
        _, st = osc.render_path(sensor, seed=0, spp=spp_cpu, max_depth=args.max_depth, threads=cores)
        el = time.perf_counter() - t0
        cpu = {"value": round(st.paths / el / 1e6, 4), "unit": "Mpaths/s", "cores": cores, "kind": "port",
               "sample": "%dx%dx%d spp of the same scene/seed (%.1f s); CPU restatement of llvm_ad_rgb (reference not installable)"
                         % (args.res, args.res, spp_cpu, el)}

    if rank == 0:
        out = {
            "metric": "Mpaths/s forward (PRB-adjoint in `prb_adjoint`), 1M-tri scene 512^2x256spp" if args.workload != "cornell" else "Mpaths/s forward path, Cornell box",
            "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%dx%dspp max_depth=%d rr_depth=5 seed=0" % (args.workload, args.res, args.res, args.spp, args.max_depth),
                       "triangles_effective": 100 * 10000 + 12 if args.workload != "cornell" else 36,
                       "accel": accel, "parallelism": "pixel-row tiles x%d, one RCCL film reduce" % world,
                       # row bands of equal measured cost (mitsuba3_amd/distributed.py BandBalancer; adapts over the first 3 frames, then frozen)
                       "row_bands": next(iter(getattr(integ, "_band_balancers", {}).values())).bounds if world > 1 and getattr(integ, "_band_balancers", None) else None},
            "prb_adjoint": prb, "roofline": roofline, "cpu_baseline": cpu,
            "stats": {k: int(v) for k, v in stats.items()},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
