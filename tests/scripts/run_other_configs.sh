# the other configurations' timings for profiles/rNN_vM_other_configs.log (run on the GPU box)
cd $GRAFT_REPO_ROOT
for w in C1 C2 C4 C5; do echo "== $w"; timeout -k 10 300 python3 tests/gpu_time_knn.py $w 2>&1 | grep stage_a | tail -1; done
echo "== scan-to-map size (tests/gpu_time_small.py: 128 particles; then 30 particles)"
timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 128 2>&1 | tail -1
timeout -k 10 200 python3 tests/gpu_time_small.py 3000 50000 30 2>&1 | tail -1
echo "== host-side phases of one bench step (tests/gpu_time_phases.py)"
for w in C3 C2 C1; do timeout -k 10 200 python3 tests/gpu_time_phases.py $w 2>&1 | tail -2; done
echo "== scan-to-map loop (tests/gpu_time_pipeline.py)"
timeout -k 10 300 python3 tests/gpu_time_pipeline.py 2>&1 | tail -4
