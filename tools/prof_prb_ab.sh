# PRB step on the textured instanced 1M scene: kernel-trace summaries for the default build (fixed-point texel accumulation) and for HAR_TQ_FIXED=0 (float LDS atomics)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
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
for t in 1 0; do export HAR_TQ_FIXED=$t;
  D=/tmp/prof_prb_f$t; rm -rf $D
  rocprofv3 --kernel-trace --stats -d $D -o r -- python /tmp/prb_only.py > gpurun_out/prof/r03_prb_fixed${t}_kt.log 2>&1
  python tools/rocpd_summary.py $(find $D -name '*.db') > gpurun_out/prof/r03_prb_fixed${t}_kt.txt 2>&1
done
head -16 gpurun_out/prof/r03_prb_fixed1_kt.txt; head -16 gpurun_out/prof/r03_prb_fixed0_kt.txt
