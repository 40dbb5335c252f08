cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_shipped
rm -rf $out; mkdir -p $out
SHIPPED=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_small.py 3000 50000 10 > $out/run.log 2>&1
head -14 $out/t/*/*kernel_stats.csv | cut -c1-160
