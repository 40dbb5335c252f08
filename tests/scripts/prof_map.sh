cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_map
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_map -o m -- python3 $R/tests/gpu_time_map.py 0.5 20 > $R/gpurun_out/prof_map.log 2>&1
f=$(find /tmp/prof_map -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/prof_map_kernel_stats.csv
tail -3 $R/gpurun_out/prof_map.log
