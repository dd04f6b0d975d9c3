# one gpurun call: full GPU suite, then a kernel trace of the bench's C4 / C5 legs (the beam-32 search dominates it)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/c5test.log 2>&1 || { tail -40 gpurun_out/c5test.log; exit 1; }; tail -12 gpurun_out/c5test.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-scoring --c4-steps 1 > gpurun_out/prof_c5.log 2>&1 || { tail -5 gpurun_out/prof_c5.log; exit 1; }
grep -o "\"c5_beam32\": {[^}]*}" gpurun_out/prof_c5.log
python3 tools/rocpd_kernels.py $(ls gpurun_out/prof_c5/*/*.db | head -1) > gpurun_out/c5_kernels.txt 2>&1; cat gpurun_out/c5_kernels.txt
