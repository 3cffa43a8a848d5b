#!/bin/bash
# Round-5 measurement on the GPU box (one gpurun call): everything tools/round_measure.sh collects, plus the BASELINE configurations (C2 - C5), the band table,
# the optimisation loops (bench.py --workload c4_loop / vertex_loop, with a kernel trace of the c4 loop) and the 8-rank rehearsal.  Outputs under gpurun_out/.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/round_measure.sh $TAG
bash tools/gpu_profile.sh $TAG ta -- --workload instanced1m
mkdir -p gpurun_out/bench gpurun_out/prof
timeout 600 python3 tools/run_configs.py > gpurun_out/bench/${TAG}_baseline_configs.jsonl 2> gpurun_out/bench/${TAG}_baseline_configs.err; cat gpurun_out/bench/${TAG}_baseline_configs.jsonl | cut -c1-200
timeout 300 python3 tools/run_configs.py --c5 >> gpurun_out/bench/${TAG}_baseline_configs.jsonl 2>> gpurun_out/bench/${TAG}_baseline_configs.err; tail -1 gpurun_out/bench/${TAG}_baseline_configs.jsonl | cut -c1-200
bash tools/bands_all.sh 8 > gpurun_out/bench/${TAG}_bands.txt 2>&1; cat gpurun_out/bench/${TAG}_bands.txt
for wl in c4_loop vertex_loop; do timeout 300 python3 bench.py --workload $wl --steps 20 --warmup 3 > gpurun_out/bench/${TAG}_$wl.json 2> gpurun_out/bench/${TAG}_$wl.err; cut -c1-300 gpurun_out/bench/${TAG}_$wl.json; done
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_c4 -o r -- python3 bench.py --workload c4_loop --steps 10 --warmup 2 --worker > gpurun_out/prof/${TAG}_c4_loop_bench.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_${TAG}_c4 -name '*.db') --json gpurun_out/prof/${TAG}_c4_loop_kt.json > gpurun_out/prof/${TAG}_c4_loop_kt.txt 2>&1
HAR_BENCH_SHARE_GPU=1 HAR_BENCH_BACKEND=gloo timeout 400 python3 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench/${TAG}_n8_rehearsal.json 2> gpurun_out/bench/${TAG}_n8_rehearsal.err; echo "n8 rehearsal rc=$?"
ls gpurun_out/bench gpurun_out/prof | grep ${TAG} | tr '\n' ' '
# round 6: kernel trace of the 1M-triangle vertex loop with the GPU-busy analysis, the single-call multi-GPU entry with 1 / 2 replicas on this GPU
bash tools/prof_vertex_loop.sh ${TAG} > gpurun_out/prof/${TAG}_vertex_loop_prof.log 2>&1; tail -3 gpurun_out/prof/${TAG}_vertex_loop_kt.txt
for k in 1 2; do HAR_BENCH_GROUP=$k timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prb --no-secondary > gpurun_out/bench/${TAG}_group$k.json 2> gpurun_out/bench/${TAG}_group$k.err; cut -c1-160 gpurun_out/bench/${TAG}_group$k.json; done
