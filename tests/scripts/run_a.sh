cd $GRAFT_REPO_ROOT
echo "== C2 default"; timeout -k 10 200 python3 tests/gpu_time_knn.py C2 2>&1 | grep stage_a | tail -1 | cut -c1-260
echo "== C1 default"; timeout -k 10 200 python3 tests/gpu_time_knn.py C1 2>&1 | grep stage_a | tail -1 | cut -c1-260
echo "== C1 knn=v1"; SVNICP_OPTIONS="knn=v1" timeout -k 10 200 python3 tests/gpu_time_knn.py C1 2>&1 | grep stage_a | tail -1 | cut -c1-260
echo "== C1 knn=v2"; SVNICP_OPTIONS="knn=v2" timeout -k 10 200 python3 tests/gpu_time_knn.py C1 2>&1 | grep stage_a | tail -1 | cut -c1-260
echo "== small default"; timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 128 2>&1 | tail -1 | cut -c1-330
echo "== small knn=v1"; SVNICP_OPTIONS="knn=v1" timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 128 2>&1 | tail -1 | cut -c1-330
echo "== small knn=v2"; SVNICP_OPTIONS="knn=v2" timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 128 2>&1 | tail -1 | cut -c1-330
for s in 8 12 16 24 32; do echo "== C3 wgpcu=$s,4"; SVNICP_OPTIONS="wgpcu=$s,4" timeout -k 10 200 python3 tests/gpu_time_knn.py C3 2>&1 | grep stage_a | tail -1 | cut -c1-200; done
