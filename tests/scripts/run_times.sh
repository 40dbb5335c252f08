# per-kernel-class GPU time of one registration for the named workloads (tests/gpu_time_knn.py), one line each
set -e
cd $GRAFT_REPO_ROOT
for w in "$@"; do timeout -k 10 200 python tests/gpu_time_knn.py $w 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_$w.log; done
