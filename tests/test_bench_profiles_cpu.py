"""The measurement contract of bench.py on the CPU: the committed rocprofv3 summaries that the bench line quotes (`roofline.traffic`, `bound_actual`, `frame_traffic`)
must describe the launch sequence of the headline frame -- one ray generation, nine closest-hit launches (the packet launch, its left-over list, seven bounces),
eight shading and eight shadow-ray launches, one splat per frame -- or bench.py quotes nothing and says "stale profile".  Round 5 shipped two sessions with a summary
in which the two flavours of k_shade (first vertex / later bounces) had collapsed into one record because tools/rocpd_summary.py cut kernel names at 60 characters."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


HEADLINE_LAUNCHES = {"raygen": 1, "trace_closest": 9, "shade": 8, "resolve": 8, "splat": 1}


def test_committed_profiles_match_the_headline_frames_launch_sequence():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test")
    timing = {k: (1.0, n) for k, n in HEADLINE_LAUNCHES.items()}
    for kind in ("traffic", "sq"):
        prof, src = bench.load_profile(kind, "instanced1m")
        assert prof is not None, "no committed %s profile for the headline workload" % kind
        assert bench.profile_matches_run(prof, timing) is True, (src, bench.profile_matches_run(prof, timing))
    prof, _ = bench.load_profile("traffic", "instanced1m")
    rec = bench.kernel_record(prof, "trace_closest")
    assert rec and {"FETCH_SIZE", "WRITE_SIZE"} <= set(rec["counters"])
    # both flavours of the shading kernel are there, under their own names
    shade = [k for k in prof if "k_shade" in k]
    assert len(shade) == 2 and len(set(shade)) == 2, shade


def test_rocpd_summary_keeps_the_template_arguments_that_tell_kernel_flavours_apart():
    rs = _load(os.path.join(ROOT, "tools", "rocpd_summary.py"), "rocpd_summary_under_test")
    a = "void har::k_shade<0, 1u, false, false, false, false, false, true, false>(har::DScene, har::ShadeParams, unsigned int)"
    b = "void har::k_shade<0, 1u, false, false, false, false, false, true, true>(har::DScene, har::ShadeParams, unsigned int)"
    assert rs.short(a) != rs.short(b)
    assert rs.short(a).endswith("true, false>") and "(" not in rs.short(a)


def _latest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_no_kernel_class_of_the_committed_run_is_modelled_above_the_hbm_peak():
    """`roofline.achieved` = algorithmic bytes / measured duration must stay below the peak for EVERY kernel class, also the ones that are not `dominant` today
    (round 5's shade branch charged 288 B per vertex: 8.3 TB/s)."""
    import json
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test")
    line = json.load(open(_latest("r0*_bench_instanced1m_driver_command.json")))
    stats, accel, ms = line["stats"], line["config"]["accel"], line["roofline"]["kernel_ms"]
    shading_bytes = 96 * accel["triangles"]
    for kernel, launches in HEADLINE_LAUNCHES.items():
        b = bench.algorithmic_bytes(kernel, stats, accel, launches, shading_bytes=shading_bytes)
        gbs = b / 1e9 / (ms[kernel] / 1e3)
        assert 0 < gbs <= bench.HBM_PEAK_GBS, (kernel, gbs)


def test_the_shade_model_is_within_1_3x_of_the_counter_measured_traffic():
    import json
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test")
    line = json.load(open(_latest("r0*_bench_instanced1m_driver_command.json")))
    prof, _ = bench.load_profile("traffic", "instanced1m")
    frames = int(bench.kernel_record(prof, "raygen")["counters"]["FETCH_SIZE"]["dispatches"])
    measured = sum((2.0 * r["counters"]["FETCH_SIZE"]["sum"] + r["counters"]["WRITE_SIZE"]["sum"]) * 1024.0 for k, r in prof.items() if "k_shade" in k) / frames
    model = bench.algorithmic_bytes("shade", line["stats"], line["config"]["accel"], HEADLINE_LAUNCHES["shade"], shading_bytes=96 * line["config"]["accel"]["triangles"])
    assert 1 / 1.3 <= model / measured <= 1.3, (model, measured)
