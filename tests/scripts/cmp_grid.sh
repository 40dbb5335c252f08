cd $GRAFT_REPO_ROOT
for wl in C2 C3 C4 C5; do for o in "X=0" "wgpcu=8,4"; do echo "== $wl $o"; if [ "$o" = "X=0" ]; then timeout -k 10 200 python tests/gpu_time_knn.py $wl 2>&1 | grep stage_a | tail -1 | cut -c1-190; else SVNICP_OPTIONS="$o" timeout -k 10 200 python tests/gpu_time_knn.py $wl 2>&1 | grep stage_a | tail -1 | cut -c1-190; fi; done; done
