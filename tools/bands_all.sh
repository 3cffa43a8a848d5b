# every 1/N band of the headline frame (forward) on one GPU, then the whole frame: what N ranks would each spend before the film reduce (usage: bands_all.sh [N=8])
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-8}
for k in $(seq 0 $((N - 1))); do python tools/band_bench.py $N 10 $k 2>&1 | tail -1 | cut -c1-70; done
python tools/band_bench.py 1 4 0 2>&1 | tail -1 | cut -c1-70
