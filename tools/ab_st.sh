#!/bin/bash
# round 5 first call: default build on the driver command (forward + PRB), then HAR_SHADING_TRIS A/B on the scenes whose geometry is not L2-resident
mkdir -p gpurun_out/ab
run() {  # name lib workload extra
  HAR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $4 > gpurun_out/ab/st_$3_$1.log 2> gpurun_out/ab/st_$3_$1.err
  python - <<PY
import json
for l in open("gpurun_out/ab/st_$3_$1.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$3 $1 fwd", j["value"], "prb", (j.get("prb_adjoint") or {}).get("value"), "kernel ms", j["roofline"]["kernel_ms"])
PY
}
BASE=$PWD/mitsuba3_amd/libhip_ad_rgb.so; ST=$PWD/tools/variants/lib_st.so
run base $BASE instanced1m
for wl in materials1m flat1m; do
  run base $BASE $wl
  run st $ST $wl
  run base2 $BASE $wl
  run st2 $ST $wl
done
run st $ST instanced1m
