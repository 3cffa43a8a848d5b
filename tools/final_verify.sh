# end-of-round verification on one GPU box: the -m gpu suite, the driver's bench command, the materials-1M bench + kernel trace, and the new kernels of the
# round (shape / normal / nested gradients, fixed-point texel accumulation) once more with every device buffer between unmapped guard pages
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r03 gpurun_out/prof
timeout 500 python -m pytest tests -m gpu -q -x > gpurun_out/r03/gpu_suite3.log 2>&1; tail -2 gpurun_out/r03/gpu_suite3.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_final3.json 2> gpurun_out/r03/bench_final3.err; tail -2 gpurun_out/r03/bench_final3.err
python bench.py --workload materials1m --steps 10 --warmup 3 --no-cpu-baseline --no-prb --no-secondary > gpurun_out/r03/bench_materials3.json 2> /dev/null
bash tools/gpu_profile.sh r03_materials1m kt -- --workload materials1m
head -8 gpurun_out/prof/r03_materials1m_kt.txt
HAR_DEBUG_GUARD=1 timeout 240 python -m pytest tests -m gpu -q -x -k "vertex_position_gradients or nested_mesh or instance_to_world_gradients or fixed_point or texel_queue or smooth_material or rough_bsdfs" > gpurun_out/r03/gpu_guard_subset.log 2>&1; tail -2 gpurun_out/r03/gpu_guard_subset.log
python - <<'PY'
import json
for f in ("gpurun_out/r03/bench_final3.json", "gpurun_out/r03/bench_materials3.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("profile_check"), (d.get("prb_adjoint") or {}).get("value"), (d.get("secondary") or {}).get("value"), d.get("cpu_baseline"))
PY
