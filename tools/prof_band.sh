cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
D=/tmp/prof_band; rm -rf $D
HAR_STREAMS=1 rocprofv3 --kernel-trace --stats -d $D -o r -- python tools/band_bench.py 8 3 > gpurun_out/prof/r03_band8_kt.log 2>&1
python tools/launch_list.py $(find $D -name '*.db') 24 > gpurun_out/prof/r03_band8_launches.txt 2>&1
cat gpurun_out/prof/r03_band8_launches.txt
tail -1 gpurun_out/prof/r03_band8_kt.log
