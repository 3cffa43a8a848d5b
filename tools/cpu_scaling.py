"""How many host cores does the box really give?  os.cpu_count(), the affinity mask, the cgroup CPU quota, and the oracle's forward rate on the benchmark
scene at several thread counts (bench.py's cpu_baseline uses the thread count this finds best)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

print("os.cpu_count", os.cpu_count(), " affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e.strerror)
try:
    print("loadavg", open("/proc/loadavg").read().strip())
    model = [l for l in open("/proc/cpuinfo") if l.startswith("model name")]
    print(len(model), "x", model[0].split(":")[1].strip())
except OSError:
    pass
sd, sensor = O.benchmark_spheres_scene(512, 512)
osc = O.OracleScene(sd)
for th in (1, 8, 16, 32, 64, 128, 256):
    spp = 1 if th < 16 else 4
    t0 = time.perf_counter(); _, st = osc.render_path(sensor, seed=0, spp=spp, max_depth=8, threads=th); el = time.perf_counter() - t0
    print("%3d threads: %.3f Mpaths/s  %.1f us x thread per path" % (th, st.paths / el / 1e6, el * th / st.paths * 1e6), flush=True)
