#!/bin/bash
# kernel trace of bench.py --workload vertex_loop (1M-triangle mesh): what the device-resident update's kernels cost per step
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_vl -o r -- python3 bench.py --workload vertex_loop --steps 10 --warmup 2 --worker > gpurun_out/prof/${TAG}_vertex_loop_bench.log 2>&1
python3 tools/rocpd_summary.py $(find /tmp/prof_${TAG}_vl -name '*.db') --busy k_develop --json gpurun_out/prof/${TAG}_vertex_loop_kt.json > gpurun_out/prof/${TAG}_vertex_loop_kt.txt 2>&1
head -70 gpurun_out/prof/${TAG}_vertex_loop_kt.txt
