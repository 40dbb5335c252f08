# development: search-kernel time with parts of the step removed (library built with -DSVNICP_DEV_ABLATE)
for a in 0 9; do
  SVNICP_ABLATE=$a timeout -k 10 120 python3 tests/gpu_time_knn.py C3 2>&1 | grep "k_stein_search" | tail -1 | sed "s/^/ablate $a: /"
done
