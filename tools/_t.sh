set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_scoring.py tests/test_gpu_c4_catalog.py tests/test_gpu_frontend.py -m gpu -q -x > gpurun_out/ref.log 2>&1 || { tail -40 gpurun_out/ref.log; exit 1; }; tail -2 gpurun_out/ref.log
python3 tools/sweep_bench.py --shapes 1000000,128,1024 1000000,128,32 1000000,128,1 1250000,256,32 --reps 20 2>&1 | grep -v amdgpu
