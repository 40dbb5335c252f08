# SQ counters of the stage-B kernels at one workload, per-launch averages (development)
cd /tmp && export TMPDIR=/tmp
w=${1:-C3}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc2_$w
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -- python3 $GRAFT_REPO_ROOT/tests/gpu_time_knn.py $w > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 $GRAFT_REPO_ROOT/tests/scripts/summarize_pmc.py $out > $out/summary.json
rm -rf $out/pmc*
