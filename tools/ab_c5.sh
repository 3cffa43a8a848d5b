#!/bin/bash
# C5 (Cornell 4096^2 x 1024 spp, 8 passes) and C2 under two settings of an environment switch, bracketed: tools/ab_c5.sh HAR_SHADE_TABLES
V=${1:-HAR_SHADE_TABLES}
for x in 1 0 1 0; do
  echo "$V=$x"; env $V=$x timeout 300 python3 tools/run_configs.py --c5 2>/dev/null | cut -c1-140
done
