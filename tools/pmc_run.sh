# PMC passes for the current kernels, one counter group per pass (never mixed with trace domains); run on the GPU box:
#   bash tools/pmc_run.sh   ->  gpurun_out/pmc_summary.csv, gpurun_out/pmc_traffic.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d gpurun_out/pmc/g1 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > gpurun_out/pmc/g1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/pmc/g2 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > gpurun_out/pmc/g2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc/g3 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > gpurun_out/pmc/g3.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc/g1 gpurun_out/pmc/g2 gpurun_out/pmc/g3 > gpurun_out/pmc_summary.csv
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc/t1 --output-format csv -- python3 tools/profile_path.py --iters 3 > gpurun_out/pmc/t1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc/t2 --output-format csv -- python3 tools/profile_path.py --iters 3 > gpurun_out/pmc/t2.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum -d gpurun_out/pmc/t3 --output-format csv -- python3 tools/profile_path.py --iters 3 > gpurun_out/pmc/t3.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc/t1 gpurun_out/pmc/t2 gpurun_out/pmc/t3 > gpurun_out/pmc_traffic.csv
rm -rf gpurun_out/pmc/*/ 
echo pmc done
