# kernel trace of the bench's C4 / C5 legs only (small C2 part): which kernels make a C5 beam step; usage: bash tools/c5_trace.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -- python3 bench.py --no-cpu-baseline --no-c3 --no-scoring --batch 64 --steps 2 --warmup 1 --c4-batch 64 --c4-steps 1 > gpurun_out/prof_c5.log 2>&1 || { tail -5 gpurun_out/prof_c5.log; exit 1; }
python3 tools/rocpd_kernels.py $(ls gpurun_out/prof_c5/*/*.db | head -1) k_beam_step > gpurun_out/c5_kernels.txt 2>&1; head -30 gpurun_out/c5_kernels.txt
