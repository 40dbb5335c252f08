# Round profile set (run on the GPU box): kernel trace + stats of bench.py, HBM traffic PMC passes and SQ PMC passes
# of one registration workload.  Outputs under gpurun_out/prof_<tag>/ ; summarise with tests/scripts/summarize_pmc.py
# and copy what should be judged into profiles/.
tag=${1:-r01}
w=${2:-C3}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $out/bench_trace.log 2>&1 || echo "trace failed" >> $out/fail.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -- python3 $root/tests/gpu_time_knn.py $w > $out/pmc$i.log 2>&1 || echo "pmc pass $i ($set) failed" >> $out/fail.log
done
cd $root && python3 tests/scripts/summarize_pmc.py $out > $out/pmc_summary.json
