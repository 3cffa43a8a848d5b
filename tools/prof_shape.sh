cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
cat > /tmp/shape_only.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import mitsuba3_amd as mi
mi.set_variant("hip_ad_rgb")
from tests.test_shape_gradients_cpu import cbox_mesh_scene, smooth_slab_scene
import os
smooth = os.environ.get("SHAPE_SCENE") == "smooth"
d = smooth_slab_scene(mi, 256, n=65) if smooth else cbox_mesh_scene(mi, 256); d["sensor"]["sampler"]["sample_count"] = 64
d["integrator"] = {"type": "prb", "max_depth": 6, "shape_gradients": [k + ".vertex_positions" for k in (("floor",) if smooth else ("small-box", "large-box", "floor"))]}
scene = mi.load_dict(d); integ = scene.integrator()
g = torch.full((256, 256, 3), 1.0 / (256 * 256 * 3), device="cuda")
for _ in range(3): integ.render_backward(scene, None, g, seed=1, spp=64)
torch.cuda.synchronize()
PY
D=/tmp/prof_shape; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o r -- python /tmp/shape_only.py > gpurun_out/prof/r03_shape_kt.log 2>&1
python tools/rocpd_summary.py $(find $D -name '*.db') > gpurun_out/prof/r03_shape_kt.txt 2>&1
head -16 gpurun_out/prof/r03_shape_kt.txt
