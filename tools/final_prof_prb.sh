cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r03 gpurun_out/prof
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
head -12 gpurun_out/prof/r03_prb_final_kt.txt
python bench.py --workload materials1m --steps 10 --warmup 3 --no-cpu-baseline --no-prb --no-secondary > gpurun_out/r03/bench_materials4.json 2> /dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_materials4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
