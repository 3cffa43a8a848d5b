for c in 16777216 33554432 67108864; do
  timeout 300 python bench.py --chunk $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/chunk_$c.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/chunk_$c.log"):
    if l.startswith("{"):
        j=json.loads(l); print("chunk $c fwd", j["value"], "ms", j["ms_per_step"], "prb", j["prb_adjoint"], j["roofline"]["kernel_ms"])
PY
  tail -2 gpurun_out/chunk_$c.log | grep -v "^{" | cut -c1-300
done
