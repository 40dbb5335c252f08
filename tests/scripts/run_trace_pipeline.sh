# rocprofv3 kernel trace of the scan-to-map loop (tests/gpu_time_pipeline.py): launches of the LAST scan
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_pipeline
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_pipeline.py > $out/run.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_\w+)", r["Kernel_Name"])
    if m: return m.group(1)
    m2 = re.search(r"(radix_sort\w*|onesweep\w*|scan\w*|histogram\w*|block_sort\w*|merge\w*|lookback\w*)", r["Kernel_Name"])
    return "rocprim:" + (m2.group(1) if m2 else r["Kernel_Name"][:50])
idx = [i for i, r in enumerate(rows) if "k_prep_crop" in r["Kernel_Name"] or "k_crop" in r["Kernel_Name"]]
if len(idx) < 2:
    idx = [i for i, r in enumerate(rows) if "k_knn_brute" in r["Kernel_Name"] or "k_knn_seed" in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]["Start_Timestamp"]); prev = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %7.1f  gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, nm(r), r.get("Grid_Size_X", "?")))
    prev = e
print("span %.1f us, %d launches" % ((int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, hi - lo))
PY
