#!/bin/bash
# the two optimisation-loop workloads, short: step time, loop / plain, time outside the library's kernels, cost of params.update()
mkdir -p gpurun_out/bench
for wl in c4_loop vertex_loop; do
  timeout 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench/r06_${wl}_b.json 2> gpurun_out/bench/r06_${wl}_b.err
  python3 - "$wl" <<'PY'
import json, sys
wl = sys.argv[1]
j = json.loads(open("gpurun_out/bench/r06_%s_b.json" % wl).read().strip().splitlines()[-1]); l = j.get("loop") or {}
print(wl, j["ms_per_step"], {k: l.get(k) for k in ("loop_over_plain", "outside_kernels_ms_per_step", "params_update_ms", "params_update_host_enqueue_ms", "params_update_no_change_ms")})
PY
done
