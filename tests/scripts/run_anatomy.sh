# development: time the search kernel of timing-only anatomy builds svn-icp_amd/libsvnicp_hip_exp<N>.so (SVNICP_SEARCH_EXPERIMENT=N, wrong results)
w=${W:-C3}
run() { SVNICP_TEST_LIB=$2 timeout -k 10 120 python3 tests/gpu_time_knn.py $w 2>&1 | grep "k_stein_search" | tail -1 | cut -c1-230 | sed "s/^/$1: /"; }
run base svn-icp_amd/libsvnicp_hip.so
for v in "$@"; do run exp$v svn-icp_amd/libsvnicp_hip_exp$v.so; done
run base svn-icp_amd/libsvnicp_hip.so
