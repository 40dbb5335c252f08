# stage-B profiling knobs on one workload: prints the plan and the per-class GPU times for each setting
cd $GRAFT_REPO_ROOT
w=${1:-C3}
run() { echo "== $*" >> gpurun_out/knobs.log; env SVNICP_OPTIONS="debug=1;$1" timeout -k 10 200 python tests/gpu_time_knn.py $w 2>&1 | grep -E "stage-B plan|stage_a" | tail -2 >> gpurun_out/knobs.log; }
: > gpurun_out/knobs.log
run X=0
run wgpcu=10,4
run wgpcu=10,2
run wgpcu=10,6
run wgpcu=8,3
run wgpcu=12,3
run wgpcu=15,3
