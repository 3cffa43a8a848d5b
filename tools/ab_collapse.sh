# A/B on the GPU: BVH collapse (dp for sets >= HAR_BVH_DP_MIN primitives | greedy)
run() {  # name dpmin workload
  HAR_BVH_DP_MIN=$2 timeout 300 python bench.py --workload $3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$3_$1.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/ab_$3_$1.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$3 $1 fwd", j["value"], "prb", (j.get("prb_adjoint") or {}).get("value"), j["roofline"]["kernel_ms"]["trace_closest"], j["roofline"]["kernel_ms"]["resolve"], j["config"]["accel"]["depth"])
PY
}
for wl in instanced1m cornell; do
  run min0 0 $wl; run min128 128 $wl; run min100000000 100000000 $wl
done
