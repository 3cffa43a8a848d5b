WL=${1:-instanced1m}
for cfg in "512 256" "512 128" "512 64" "512 32" "256 32"; do set -- $cfg
  timeout 300 python bench.py --workload $WL --res $1 --spp $2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/spp_$1_$2.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/spp_$1_$2.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$WL res $1 spp $2 fwd", j["value"], "ms", j["ms_per_step"], "prb", j["prb_adjoint"]["value"], j["roofline"]["kernel_ms"])
PY
done
