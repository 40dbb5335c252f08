cd $GRAFT_REPO_ROOT
for s in 2 4 6 8 12 16 24; do echo "== C2 wgpcu=$s,4"; SVNICP_OPTIONS="wgpcu=$s,4" timeout -k 10 200 python3 tests/gpu_time_knn.py C2 2>&1 | grep stage_a | tail -1 | cut -c1-200; done
