set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vs_oracle" > gpurun_out/t_multi.log 2>&1
for w in P1024 C4 P256 C3; do timeout -k 10 200 python tests/gpu_time_knn.py $w 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_$w.log; done
SVNICP_OPTIONS="fused_update_max_p=100" timeout -k 10 200 python tests/gpu_time_knn.py P256 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_P256_multi.log
SVNICP_OPTIONS="fused_update_max_p=100" timeout -k 10 200 python tests/gpu_time_knn.py C3 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_C3_multi.log
SVNICP_OPTIONS="fused_update_max_p=20" timeout -k 10 200 python tests/gpu_time_knn.py C2 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_C2_multi.log
timeout -k 10 200 python tests/gpu_time_knn.py C2 2>&1 | grep stage_a | tail -1 > gpurun_out/tm_C2.log
