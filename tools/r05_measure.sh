# one gpurun call: GPU tests, bench line, kernel trace of the bench command, the three PMC passes of `bench.py --pmc-run`
# (SQ counters, FETCH_SIZE, WRITE_SIZE: separate passes, program directly after `--`), PMC pass of the sweeps.
# usage: bash tools/r05_measure.sh <tag> [notest|nopmc]; every step stops the script when it fails.
set -o pipefail
V=${1:-r05_v1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" != "notest" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gputest_${V}.log 2>&1 || { tail -40 gpurun_out/gputest_${V}.log; exit 1; }; tail -4 gpurun_out/gputest_${V}.log
fi
timeout -k 10 600 python bench.py > gpurun_out/bench_${V}.json 2> gpurun_out/bench_${V}.err || { tail -5 gpurun_out/bench_${V}.err; exit 1; }
python3 tools/bench_summary.py gpurun_out/bench_${V}.json
if [ "$2" == "nopmc" ]; then exit 0; fi
python3 tools/sweep_bench.py --shapes 1000000,128,1024 1000000,128,32 1000000,128,1 1250000,256,1024 1250000,256,32 1250000,256,1 3415,128,1024 --reps 20 > gpurun_out/sweep_bench_${V}.txt 2>&1 || { tail -5 gpurun_out/sweep_bench_${V}.txt; exit 1; }; cat gpurun_out/sweep_bench_${V}.txt
rm -rf /tmp/irs_prof; mkdir -p /tmp/irs_prof
rocprofv3 --kernel-trace --stats -d /tmp/irs_prof/prof_bench -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/prof_bench.log 2>&1 || { tail -5 gpurun_out/prof_bench.log; exit 1; }
python3 tools/rocpd_kernels.py $(ls /tmp/irs_prof/prof_bench/*/*.db | head -1) k_path_step > gpurun_out/c2_b4096_kernels_${V}.txt 2>&1; head -14 gpurun_out/c2_b4096_kernels_${V}.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/irs_prof/pmc_c2_sq -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_c2_rows.json 2> gpurun_out/pmc_c2_sq.err || { tail -5 gpurun_out/pmc_c2_sq.err; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/irs_prof/pmc_c2_fetch -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_c2_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_c2_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/irs_prof/pmc_c2_write -- python3 bench.py --pmc-run --steps 5 --warmup 2 > gpurun_out/pmc_c2_write.log 2>&1 || { tail -5 gpurun_out/pmc_c2_write.log; exit 1; }
python3 tools/r05_pmc.py /tmp/irs_prof/pmc_c2_sq /tmp/irs_prof/pmc_c2_fetch /tmp/irs_prof/pmc_c2_write gpurun_out/pmc_c2_rows.json gpurun_out/c2_b4096_pmc_${V}.json || exit 1
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
