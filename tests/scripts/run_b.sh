cd $GRAFT_REPO_ROOT
echo "== C1 default"; timeout -k 10 200 python3 tests/gpu_time_knn.py C1 2>&1 | grep stage_a | tail -1 | cut -c1-260
echo "== small 128"; timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 128 2>&1 | tail -1 | cut -c1-330
echo "== small 30"; timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 30 2>&1 | tail -1 | cut -c1-330
echo "== C2"; timeout -k 10 200 python3 tests/gpu_time_knn.py C2 2>&1 | grep stage_a | tail -1 | cut -c1-260
