# rocprofv3 kernel trace of a small registration (tests/gpu_time_small.py) and of C1: timeline of the last registration
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_small
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/p128 -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_small.py 3000 50000 128 > $out/p128.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/c1 -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_phases.py C1 > $out/c1.log 2>&1
python3 - <<PY
import csv, glob
for tag in ("p128", "c1"):
    f = glob.glob("$out/%s/**/*kernel_trace.csv" % tag, recursive=True)
    if not f: print(tag, "no trace"); continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    # last registration = from the last k_bbox on (C1 / tiles) or the last 140 kernels
    idx = [i for i, r in enumerate(rows) if "k_bbox" in r["Kernel_Name"] or "k_targets_soa" in r["Kernel_Name"]]
    lo = idx[-2] if len(idx) > 1 else max(0, len(rows) - 140)
    hi = idx[-1] if len(idx) > 1 else len(rows)
    t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
    print("==", tag, "kernels", hi - lo)
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-60:]
        print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
        prev_end = e
PY
