# development: time stage B for prebuilt library variants svn-icp_amd/libsvnicp_hip_<tag>.so (first: the current library)
timeout -k 10 120 python3 tests/gpu_time_knn.py C3 2>&1 | grep "k_stein_search" | tail -1 | cut -c40-200 | sed "s/^/current: /"
cp svn-icp_amd/libsvnicp_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp svn-icp_amd/libsvnicp_hip_$v.so svn-icp_amd/libsvnicp_hip.so
  timeout -k 10 120 python3 tests/gpu_time_knn.py C3 2>&1 | grep "k_stein_search" | tail -1 | cut -c40-200 | sed "s/^/$v: /"
done
cp /tmp/lib_keep.so svn-icp_amd/libsvnicp_hip.so
