# SQ counters for the stage-B kernel of one workload (one pass per counter set; summary per kernel name)
cd /tmp && export TMPDIR=/tmp
w=${1:-C3}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$w
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_knn.py $w > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; k = "search" if "k_stein_search" in k else "accum_w" if "k_stein_accumulate" in k else "knn_tiles" if "k_knn_tiles" in k else None
        if k is None: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in acc.items():
        if True:
            print(f.split("/")[-3], k, {c: v for c, v in d.items()})
PY
