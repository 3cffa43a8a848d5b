#!/bin/bash
# the driver's bench commands, literally (N = 1) and the self-launching N = 2 rehearsal on one shared GPU (gloo: never a reported number)
mkdir -p gpurun_out/bench
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench/n1.out 2> gpurun_out/bench/n1.err; echo "n1 rc=$?"; cat gpurun_out/bench/n1.err | grep -v amdgpu.ids | tail -20; cut -c1-3000 gpurun_out/bench/n1.out
HAR_BENCH_SHARE_GPU=1 HAR_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench/n2_rehearsal.out 2> gpurun_out/bench/n2_rehearsal.err; echo "n2 rc=$?"; tail -5 gpurun_out/bench/n2_rehearsal.err; cut -c1-600 gpurun_out/bench/n2_rehearsal.out
