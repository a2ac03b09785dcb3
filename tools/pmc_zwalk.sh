# PMC passes for the z-walk kernel alone (one counter group per pass, never mixed with trace domains):
#   bash tools/pmc_zwalk.sh [variant] -> gpurun_out/pmc_zwalk_v<variant>.csv
set -e
V=${1:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcz; mkdir -p $O
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/g$i --output-format csv -- python3 tools/profile_zwalk.py --variant $V > $O/g$i.log 2>&1 || echo "group $i failed"
done
python3 tools/pmc_summary.py $O/g* > gpurun_out/pmc_zwalk_v$V.csv
rm -rf $O
grep -E "i8z|kernel," gpurun_out/pmc_zwalk_v$V.csv
