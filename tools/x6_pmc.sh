# PMC passes over the x6 lab (k_block<f32> vs k_block_x6); usage: bash tools/x6_pmc.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-a}
rm -rf gpurun_out/pmc_x6a gpurun_out/pmc_x6b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_x6a -- tools/x6_lab 131072 0 > gpurun_out/pmc_x6a.log 2>&1 || { tail -5 gpurun_out/pmc_x6a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_x6b -- tools/x6_lab 131072 0 > gpurun_out/pmc_x6b.log 2>&1 || { tail -5 gpurun_out/pmc_x6b.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_x6a k_block > gpurun_out/x6_pmc_${V}.txt
python3 tools/pmc_summary.py gpurun_out/pmc_x6b k_block >> gpurun_out/x6_pmc_${V}.txt
cat gpurun_out/x6_pmc_${V}.txt
