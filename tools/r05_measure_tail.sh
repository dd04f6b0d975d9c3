# the second half of tools/r05_measure.sh alone (sweep PMC pass, the two-kernel decoder path for comparison, the size sweep)
set -o pipefail
V=${1:-r05_v1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/irs_prof
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/irs_prof/pmc_sweep3 -- python3 tools/sweep_bench.py --shapes 1000000,128,1024 1250000,256,1024 --reps 3 > gpurun_out/pmc_run.log 2>&1 || { tail -5 gpurun_out/pmc_run.log; exit 1; }
python3 tools/pmc_summary.py /tmp/irs_prof/pmc_sweep3 k_sweep k_select k_refine k_prep > gpurun_out/sweep_pmc_${V}.txt; cat gpurun_out/sweep_pmc_${V}.txt
# ---- the two-kernel decoder path for comparison (IRS_DECODER_SEQ=0): bench line, kernel trace, the three PMC passes
export IRS_DECODER_SEQ=0
timeout -k 10 400 python bench.py --no-scoring --no-c4 --no-latency --no-cpu-baseline > gpurun_out/bench_twokernel_${V}.json 2> gpurun_out/bench_twokernel_${V}.err || { tail -5 gpurun_out/bench_twokernel_${V}.err; exit 1; }
python3 tools/bench_summary.py gpurun_out/bench_twokernel_${V}.json | head -4
rm -rf /tmp/irs_prof/prof_seq /tmp/irs_prof/pmc_seq_sq /tmp/irs_prof/pmc_seq_fetch /tmp/irs_prof/pmc_seq_write
rocprofv3 --kernel-trace --stats -d /tmp/irs_prof/prof_seq -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/prof_seq.log 2>&1 || { tail -5 gpurun_out/prof_seq.log; exit 1; }
python3 tools/rocpd_kernels.py $(ls /tmp/irs_prof/prof_seq/*/*.db | head -1) k_path_step > gpurun_out/c2_b4096_kernels_twokernel_${V}.txt 2>&1; head -12 gpurun_out/c2_b4096_kernels_twokernel_${V}.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/irs_prof/pmc_seq_sq -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_seq_rows.json 2> gpurun_out/pmc_seq_sq.err || { tail -5 gpurun_out/pmc_seq_sq.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/irs_prof/pmc_seq_fetch -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_seq_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_seq_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/irs_prof/pmc_seq_write -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_seq_write.log 2>&1 || { tail -5 gpurun_out/pmc_seq_write.log; exit 1; }
python3 tools/r05_pmc.py /tmp/irs_prof/pmc_seq_sq /tmp/irs_prof/pmc_seq_fetch /tmp/irs_prof/pmc_seq_write gpurun_out/pmc_seq_rows.json gpurun_out/c2_b4096_pmc_twokernel_${V}.json || exit 1
unset IRS_DECODER_SEQ
# ---- size sweep of the 6-layer decode alone: where the sequence-resident launch overtakes the two-kernel path
for n in 128 256 384 512 768 1024 1536 2048 4096; do timeout -k 10 120 python tools/seq_probe.py $n 6 2>/dev/null | head -2; done > gpurun_out/seq_sizes_${V}.txt 2>&1; cat gpurun_out/seq_sizes_${V}.txt
