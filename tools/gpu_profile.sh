#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of `python bench.py $BENCH_ARGS`,
# summaries written to gpurun_out/prof/<tag>_*.txt (copy the ones to keep into profiles/).
# every rocprofv3 call runs under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then wait forever (cost 15 GPU-minutes once)
# usage: tools/gpu_profile.sh <tag> [kt] [sq] [sq2] [ic] [ta] [mem] [tcc] -- <bench args>
set -u
TAG=$1; shift
PASSES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
[ $# -gt 0 ] && shift
ARGS="$* --no-cpu-baseline --no-prb --no-secondary --worker"      # --worker: measure in THIS process (bench.py without it is a launcher; rocprofv3 must see the kernels)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
for P in "${PASSES[@]}"; do
  D=/tmp/prof_${TAG}_$P; rm -rf $D
  case $P in
    kt)  timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats -d $D -o r -- python bench.py --steps 2 --warmup 1 $ARGS > $OUT/${TAG}_kt_bench.log 2>&1 ;;
    sq)  timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > $OUT/${TAG}_sq_bench.log 2>&1 ;;
    mem) timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc FETCH_SIZE -d ${D}f -o r -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2>&1
         timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc WRITE_SIZE -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2>&1
         python tools/rocpd_summary.py $(find ${D}f -name '*.db') $(find $D -name '*.db') --json $OUT/${TAG}_traffic.json > $OUT/${TAG}_fetch_write.txt 2>&1 ;;
    sq2) timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > $OUT/${TAG}_sq2_bench.log 2>&1 ;;
    ic)  timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > $OUT/${TAG}_ic_bench.log 2>&1 ;;
    ta)  timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > $OUT/${TAG}_ta_bench.log 2>&1 ;;      # more TA counters in one pass exceed the hardware: rocprofv3 aborts, then hangs in its finaliser
    tcc) timeout -k 5 ${PROF_TIMEOUT:-240} rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $D -o r -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2>&1 ;;
  esac
  LAUNCHES=""; [ "$P" = kt ] && LAUNCHES="--launches $OUT/${TAG}_launches.txt"
  python tools/rocpd_summary.py $(find $D -name '*.db') --json $OUT/${TAG}_$P.json $LAUNCHES > $OUT/${TAG}_$P.txt 2>&1
  [ "$P" = kt ] && find $D -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
done
