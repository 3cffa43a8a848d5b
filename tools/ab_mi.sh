#!/bin/bash
# HAR_HIT_MATINFO A/B: default build (1) against -DHAR_HIT_MATINFO=0, bracketed, materials1m (generic shading kernel) and instanced1m (diffuse kernels, forward + PRB)
mkdir -p gpurun_out/ab
run() {  # name lib workload extra
  HAR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $4 > gpurun_out/ab/mi_$3_$1.log 2> gpurun_out/ab/mi_$3_$1.err
  python - <<PY
import json
for l in open("gpurun_out/ab/mi_$3_$1.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$3 $1 fwd", j["value"], "prb", (j.get("prb_adjoint") or {}).get("value"), "kernel ms", j["roofline"]["kernel_ms"])
PY
}
NEW=$PWD/mitsuba3_amd/libhip_ad_rgb.so; OLD=$PWD/tools/variants/lib_nomi.so
for wl in materials1m instanced1m; do
  run matinfo $NEW $wl
  run plain $OLD $wl
  run matinfo2 $NEW $wl
  run plain2 $OLD $wl
done
