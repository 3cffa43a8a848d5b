#!/bin/bash
# quick A/B on the GPU box: full GPU test suite once, then bench lines (forward + PRB, no CPU baseline) for the listed env variants
# usage: tools/ab_bench.sh "<tag>:<ENV=VAL ...>" ...
mkdir -p gpurun_out/ab
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 900 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3; fi
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ab/$tag.out 2> gpurun_out/ab/$tag.err
  python3 - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab/$tag.out").read().strip().splitlines()[-1])
    r = j["roofline"]; p = j.get("prb_adjoint") or {}
    print("$tag: fwd %.1f Mpaths/s (%.2f ms)  prb %s  kernels %s  trace avg %.3f ms" % (j["value"], j["ms_per_step"], p.get("value"), r["kernel_ms"], r["avg_launch_ms"]))
except Exception as e:
    print("$tag: FAILED", e, open("gpurun_out/ab/$tag.err").read()[-400:])
PY
done
