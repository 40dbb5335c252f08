# stage-B profiling knobs on one workload: prints the per-class GPU times for each setting (search, accumulate workgroups per CU)
cd $GRAFT_REPO_ROOT
w=${1:-C3}
run() { echo "== $*" >> gpurun_out/knobs.log; env SVNICP_OPTIONS="$1" timeout -k 10 200 python tests/gpu_time_knn.py $w 2>&1 | grep -E "stage_a" | tail -1 | cut -c1-200 >> gpurun_out/knobs.log; }
: > gpurun_out/knobs.log
for s in 16 24 32 48 64; do run "wgpcu=$s,4"; done

cat gpurun_out/knobs.log
