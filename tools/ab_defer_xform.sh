# A/B on one box, instanced1m forward + PRB: base (ray transitions where they happen), dx (HAR_DEFER_XFORM=1: one modification site), dx6 (the same + 6-wave launch bound)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03
OUT=gpurun_out/r03/ab_defer_xform.txt; : > $OUT
for rep in 1 2; do
  for v in base dx dx6; do
    if [ $v = dx ]; then LIB=mitsuba3_amd/libhip_ad_rgb.so; else LIB=tools/variants/lib_$v.so; fi
    HAR_LIB_PATH=$PWD/$LIB python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], 'Mpaths/s', d['ms_per_step'], 'ms  prb', d['prb_adjoint']['value'], d['roofline']['kernel_ms'])" >> $OUT
  done
done
cat $OUT
