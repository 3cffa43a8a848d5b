cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r03 gpurun_out/prof
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
tail -3 gpurun_out/r03/bench_final.err
bash tools/gpu_profile.sh r03_instanced1m kt mem sq -- --workload instanced1m
# PRB step (record tape, default build)
cat > /tmp/prb_only.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
d = mi.instanced_spheres_scene(width=512, height=512, spp=256, textured=True)
d["integrator"] = {"type": "prb", "max_depth": 8, "rr_depth": 5, "emitter_gradients": True}
scene = mi.load_dict(d); integ = scene.integrator()
g = torch.full((512, 512, 3), 1.0 / (512 * 512 * 3), device="cuda")
for _ in range(3):
    mi.render_backward_distributed(scene, g, integ, seed=1, spp=256)
torch.cuda.synchronize()
PY
D=/tmp/prof_prb_final; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o r -- python /tmp/prb_only.py > gpurun_out/prof/r03_prb_final_kt.log 2>&1
python tools/rocpd_summary.py $(find $D -name '*.db') > gpurun_out/prof/r03_prb_final_kt.txt 2>&1
head -16 gpurun_out/prof/r03_prb_final_kt.txt
head -12 gpurun_out/prof/r03_instanced1m_kt.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_final2.json 2> gpurun_out/r03/bench_final2.err
python -c "
import json
for f in ('gpurun_out/r03/bench_final.json','gpurun_out/r03/bench_final2.json'):
    d=json.load(open(f)); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["profile_check"], (d.get("prb_adjoint") or {}).get("value"), (d.get("secondary") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
"
