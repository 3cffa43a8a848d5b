for leaf in 1 2 3; do for ct in 0.3 0.15; do for wl in instanced1m flat1m; do
  HAR_BLAS_MAX_LEAF=$leaf HAR_BVH_CTRI=$ct timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-prb > gpurun_out/leaf.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/leaf.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$wl leaf $leaf ctri $ct fwd", j["value"], j["roofline"]["kernel_ms"]["trace_closest"], j["roofline"]["kernel_ms"]["resolve"], j["config"]["accel"]["nodes"], j["config"]["accel"]["depth"])
PY
done; done; done
