#!/bin/bash
# A/B: two staggered half-frames on two streams (HAR_STREAMS=2 HAR_DUAL_STAGGER=1) with smaller persistent grids (HAR_TRACE_GRID), against the one-stream default
SKIP_TESTS=1 BENCH_ARGS="--no-secondary" bash tools/ab_bench.sh \
  "off:HAR_X=0" \
  "s2:HAR_STREAMS=2" \
  "s2_stag:HAR_STREAMS=2 HAR_DUAL_STAGGER=1" \
  "s2_stag_1280:HAR_STREAMS=2 HAR_DUAL_STAGGER=1 HAR_TRACE_GRID=1280" \
  "s2_stag_1024:HAR_STREAMS=2 HAR_DUAL_STAGGER=1 HAR_TRACE_GRID=1024" \
  "s2_1280:HAR_STREAMS=2 HAR_TRACE_GRID=1280" \
  "g1280:HAR_TRACE_GRID=1280" \
  "g1536:HAR_TRACE_GRID=1536" \
  "off2:HAR_X=0"
