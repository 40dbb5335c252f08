# rocprofv3 kernel trace + stats of one workload through tests/gpu_time_knn.py
cd /tmp && export TMPDIR=/tmp
w=${1:-C3}
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$w
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_knn.py $w > $out.log 2>&1
