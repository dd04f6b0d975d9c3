# SQ counters of the decoder kernels in one C2 step (bench.py --pmc-run); usage: bash tools/attn_pmc.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-a}
rm -rf gpurun_out/pmc_at_a gpurun_out/pmc_at_b
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_at_a -- python3 bench.py --pmc-run --steps 3 --warmup 1 > gpurun_out/pmc_at_a.log 2>&1 || { tail -5 gpurun_out/pmc_at_a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_at_b -- python3 bench.py --pmc-run --steps 3 --warmup 1 > gpurun_out/pmc_at_b.log 2>&1 || { tail -5 gpurun_out/pmc_at_b.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_at_a k_attn k_block k_embed > gpurun_out/attn_pmc_${V}.txt
python3 tools/pmc_summary.py gpurun_out/pmc_at_b k_attn k_block k_embed >> gpurun_out/attn_pmc_${V}.txt
cut -c1-330 gpurun_out/attn_pmc_${V}.txt
