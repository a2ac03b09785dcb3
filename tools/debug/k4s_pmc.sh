# counters of the sparse correlation kernels: bash tools/debug/k4s_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/k4pmc; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $O/a --output-format csv -- python3 tools/debug/k4s_phases.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY -d $O/b --output-format csv -- python3 tools/debug/k4s_phases.py > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU -d $O/c --output-format csv -- python3 tools/debug/k4s_phases.py > $O/c.log 2>&1
python3 tools/pmc_summary.py $O/a $O/b $O/c | grep -E "corr_gather|corr_lists|kernel,counter"
rm -rf $O/a $O/b $O/c
