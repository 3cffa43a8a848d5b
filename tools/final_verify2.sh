cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r03 gpurun_out/prof
timeout 300 python -m pytest tests -m gpu -q > gpurun_out/r03/gpu_suite6.log 2>&1; tail -2 gpurun_out/r03/gpu_suite6.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_final4.json 2> gpurun_out/r03/bench_final4.err; tail -1 gpurun_out/r03/bench_final4.err
bash tools/gpu_profile.sh r03_instanced1m kt mem sq -- --workload instanced1m
head -6 gpurun_out/prof/r03_instanced1m_kt.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03/bench_final4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("profile_check"), (d.get("prb_adjoint") or {}).get("value"), (d.get("secondary") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
PY
