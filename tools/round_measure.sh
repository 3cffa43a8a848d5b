timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_instanced1m.log 2> gpurun_out/bench_instanced1m.err; tail -1 gpurun_out/bench_instanced1m.log | cut -c1-600
bash tools/gpu_profile.sh r01g kt sq mem -- --workload instanced1m
for wl in flat1m cornell materials1m; do timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$wl.log 2>&1; tail -1 gpurun_out/bench_$wl.log | cut -c1-200; done
ls gpurun_out/prof | tail
