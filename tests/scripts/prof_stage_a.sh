#!/bin/bash
# per-kernel times of stage A at C3 and C5 (run on the GPU box): tests/scripts/prof_stage_a.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in C3 C5; do
  rm -rf /tmp/prof_sa_$wl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sa_$wl -o sa -- python3 $R/tests/gpu_knn_heavy.py $wl > $R/gpurun_out/prof_sa_$wl.log 2>&1 || exit 1
  f=$(find /tmp/prof_sa_$wl -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/prof_sa_${wl}_kernel_stats.csv
  echo "== $wl"; head -14 "$f" | cut -d, -f1-4
done
