#!/bin/bash
# Round measurement on the GPU box (one gpurun call): the GPU suite, the driver's literal bench command, rocprofv3 passes of the same workload
# (kernel trace + stats, SQ counters, FETCH / WRITE in separate --pmc runs), the other workloads, the N = 2 rehearsal.  Outputs under gpurun_out/.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/bench gpurun_out/prof
timeout 900 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench/${TAG}_n1.json 2> gpurun_out/bench/${TAG}_n1.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench/${TAG}_n1.json
bash tools/gpu_profile.sh ${TAG} kt sq mem -- --workload instanced1m
for wl in flat1m cornell materials1m; do timeout 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench/${TAG}_$wl.json 2> gpurun_out/bench/${TAG}_$wl.err; cut -c1-200 gpurun_out/bench/${TAG}_$wl.json; done
HAR_BENCH_SHARE_GPU=1 HAR_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench/${TAG}_n2_rehearsal.json 2> gpurun_out/bench/${TAG}_n2_rehearsal.err; echo "n2 rehearsal rc=$?"
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_prb -o r -- python3 tools/prb_breakdown.py instanced1m textured > gpurun_out/prof/${TAG}_prb_bench.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_${TAG}_prb -name '*.db') --json gpurun_out/prof/${TAG}_prb_kt.json > gpurun_out/prof/${TAG}_prb_kt.txt 2>&1
ls gpurun_out/prof | grep ${TAG}
