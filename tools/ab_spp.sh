WL=${1:-instanced1m}
for cfg in "512 256" "1024 64" "2048 16" "4096 4"; do set -- $cfg
  timeout 300 python bench.py --workload $WL --res $1 --spp $2 --steps 2 --warmup 1 --no-cpu-baseline --no-prb > gpurun_out/spp_$1_$2.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/spp_$1_$2.log"):
    if l.startswith("{"):
        j=json.loads(l); print("$WL res $1 spp $2 fwd", j["value"], j["roofline"]["kernel_ms"], j["stats"])
PY
done
