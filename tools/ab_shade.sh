run() {  # name lib workload
  HAR_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abs_$3_$1.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/abs_$3_$1.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$3 $1 fwd", j["value"], "prb", (j.get("prb_adjoint") or {}).get("value"), "shade ms", j["roofline"]["kernel_ms"]["shade"])
PY
}
for wl in instanced1m cornell; do
  run base $PWD/mitsuba3_amd/libhip_ad_rgb.so $wl
  run sw5 $PWD/tools/variants/lib_sw5.so $wl
  run sw6 $PWD/tools/variants/lib_sw6.so $wl
  run sw8 $PWD/tools/variants/lib_sw8.so $wl
done
