set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_frontend.py -m gpu -q -x > gpurun_out/traintest.log 2>&1 || { tail -60 gpurun_out/traintest.log; exit 1; }; tail -3 gpurun_out/traintest.log
