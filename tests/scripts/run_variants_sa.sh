# development: time stage A for prebuilt library variants svn-icp_amd/libsvnicp_hip_<tag>.so (first: the current library)
cd $GRAFT_REPO_ROOT
for wl in C3 C5; do timeout -k 10 200 python3 tests/gpu_knn_heavy.py $wl 2>&1 | grep "stage A" | tail -1 | sed "s/^/current $wl: /"; done
cp svn-icp_amd/libsvnicp_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp svn-icp_amd/libsvnicp_hip_$v.so svn-icp_amd/libsvnicp_hip.so
  for wl in C3 C5; do timeout -k 10 200 python3 tests/gpu_knn_heavy.py $wl 2>&1 | grep "stage A" | tail -1 | sed "s/^/$v $wl: /"; done
done
cp /tmp/lib_keep.so svn-icp_amd/libsvnicp_hip.so
