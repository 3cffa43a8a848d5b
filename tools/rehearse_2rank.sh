cd ${GRAFT_REPO_ROOT:-.}
for cfg in "HAR_OVERLAP=-1" "HAR_OVERLAP=0" "HAR_TORCH_ALLOCATOR=0" "HAR_OVERLAP=0 HAR_TORCH_ALLOCATOR=0 HAR_STREAMS=1"; do
  echo "== $cfg"
  env $(echo $cfg | sed 's/HAR_OVERLAP=-1//') HAR_BENCH_SHARE_GPU=1 HAR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep "forward done\|PRB adjoint done"
done
