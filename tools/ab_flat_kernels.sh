# A/B on one box: scenes without a TLAS with the FLAT instantiation of the traversal kernels (default) against the generic one (HAR_FLAT_KERNELS=0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/ab_flat_kernels.txt; : > $OUT
for wl in flat1m materials1m cornell; do
  for fk in 0 1 0 1; do
    HAR_FLAT_KERNELS=$fk python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-prb --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'HAR_FLAT_KERNELS=$fk', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms', d['roofline']['kernel_ms'], d['stats'])" >> $OUT
  done
done
cat $OUT
